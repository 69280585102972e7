"""How long does pinning the caller's frame block take, against copying it pageable?"""
import time, numpy as np, torch
a = np.random.randint(0, 255, size=(256, 512, 512), dtype=np.uint8)
d = torch.empty(a.shape, dtype=torch.uint8, device='cuda')
t = torch.from_numpy(a)
rt = torch.cuda.cudart()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.copy_(t); torch.cuda.synchronize(); t1 = time.perf_counter()
    r = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); t2 = time.perf_counter()
    d.copy_(t, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    rt.cudaHostUnregister(a.ctypes.data); t4 = time.perf_counter()
    print('pageable copy %.2f ms | register %.2f ms (rc %s) copy %.2f ms unregister %.2f ms' % (
        1e3 * (t1 - t0), 1e3 * (t2 - t1), r, 1e3 * (t3 - t2), 1e3 * (t4 - t3)))
