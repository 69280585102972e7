"""The C oracle under AddressSanitizer + UBSan (CPU build only; GPU ASan is not
available on the pool).  Runs a few fixtures, including failure paths, in a
subprocess with the sanitizer runtime preloaded."""
import os
import subprocess
import sys

import pytest

import _cases

SCRIPT = r'''
import sys, ctypes
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tests'); sys.path.insert(0, %(root)r + '/oracle')
import ctr_oracle
ctr_oracle.LIB_PATH = %(lib)r
import _cases
for name in ('cfg1_triple', 'edges', 'oob_feature', 'nan_feature', 'dimer_constrained',
             'tetramer3d_constrained', 'aniso3d_sizevar', 'rms_threshold', 'dtype_u16'):
    case = _cases.Case(name)
    res = case.run(lambda p, b: ctr_oracle.run_batch(p, b, 2))
    print(name, len(res))
print('SANITIZED-OK')
'''


def test_oracle_under_asan_ubsan(tmp_path):
    oracle_dir = os.path.join(_cases.ROOT, 'oracle')
    subprocess.check_call(['make', '-s', '-C', oracle_dir, 'asan'])
    lib = os.path.join(oracle_dir, '_build', 'libctr_oracle_asan.so')
    asan_rt = subprocess.check_output(['gcc', '-print-file-name=libasan.so']).decode().strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("libasan runtime not found")
    env = dict(os.environ, LD_PRELOAD=asan_rt,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1')
    out = subprocess.run([sys.executable, '-c', SCRIPT % dict(root=_cases.ROOT, lib=lib)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'SANITIZED-OK' in out.stdout
    assert 'runtime error' not in out.stderr and 'AddressSanitizer' not in out.stderr
