"""Device-resident batches: inputs live in HBM (torch tensors are used only as
device memory + stream plumbing), the engine runs asynchronously on a HIP
stream through ``ctr_refine_batch_device`` (include/ctrefine.h)."""
import numpy as np

from . import _abi, _lib


class DeviceBatch(object):
    """A HostBatch uploaded once; ``run()`` queues one pass of the hot path."""

    def __init__(self, problem, host_batch, device=0, engine=None, result_rows=0, done_flag=None):
        """result_rows: > 0 allocates ``t['result_rows']`` with that many rows (>= the features
        of the batch; rows beyond them stay zero) of n_params + 1 columns and has the engine write
        params_out | cost-of-the-row's-cluster into it (``ctr_batch.result_rows``): the block a
        multi-GPU pipeline gathers, padded to the same row count on every rank.  A float64 CUDA
        tensor instead of a count is used as that block as it is -- e.g. this rank's part of
        rank 0's inbox, so that the rows reach rank 0 as they are written; ``(address, rows)``
        names such a block by its address (memory mapped with ``Engine.ipc_open``).
        done_flag: a one-element int64 CUDA tensor, or an address, into which every call stores
        ``done_value`` (set ``struct.done_value`` before ``run``) when the batch is finished."""
        import torch
        self.torch = torch
        self.device = torch.device('cuda', device)
        self.engine = engine or _lib.default_engine(device)
        self.problem = problem
        self.host = host_batch
        hb = host_batch
        frames = hb.frames
        if frames.dtype == np.uint16:   # torch has no uint16 arithmetic; bytes are what travels
            frames_t = torch.from_numpy(frames.view(np.int16))
        else:
            frames_t = torch.from_numpy(frames)
        with torch.cuda.device(self.device):
            self.t = dict(
                frames=frames_t.to(self.device),
                frame_index=torch.from_numpy(hb.frame_index).to(self.device),
                feat_offset=torch.from_numpy(hb.feat_offset).to(self.device),
                params=torch.from_numpy(hb.params).to(self.device),
                low=torch.from_numpy(hb.low).to(self.device),
                high=torch.from_numpy(hb.high).to(self.device),
                params_out=torch.empty(hb.params.shape, dtype=torch.float64, device=self.device),
                cost=torch.empty(hb.n_clusters, dtype=torch.float64, device=self.device),
                status=torch.empty(hb.n_clusters, dtype=torch.int32, device=self.device),
                n_rounds=torch.empty(hb.n_clusters, dtype=torch.int32, device=self.device),
                n_iter=torch.empty(hb.n_clusters, dtype=torch.int32, device=self.device),
            )
            torch.cuda.synchronize(self.device)
        if hb.params_std is not None:
            with torch.cuda.device(self.device):
                self.t['params_std'] = torch.empty(hb.params.shape, dtype=torch.float64, device=self.device)
        self.raw_rows = None
        if isinstance(result_rows, tuple):
            # (address, rows): a block this process has mapped (ctr_ipc_open) or owns (ctr_ipc_alloc)
            addr, n_rows = result_rows
            if int(n_rows) < hb.n_features:
                raise ValueError("result_rows must hold every feature of the batch")
            self.raw_rows = int(addr)
        elif hasattr(result_rows, 'data_ptr'):
            ext = result_rows
            if ext.dtype != torch.float64 or not ext.is_contiguous() or ext.dim() != 2 or \
                    ext.shape[0] < hb.n_features or ext.shape[1] != hb.params.shape[1] + 1:
                raise ValueError("result_rows tensor must be contiguous float64 [>= n_features, n_params + 1]")
            self.t['result_rows'] = ext
        elif result_rows:
            if result_rows < hb.n_features:
                raise ValueError("result_rows must hold every feature of the batch")
            with torch.cuda.device(self.device):
                self.t['result_rows'] = torch.zeros((int(result_rows), hb.params.shape[1] + 1),
                                                    dtype=torch.float64, device=self.device)
        b = hb.as_struct()
        for name, tensor in self.t.items():
            setattr(b, name, tensor.data_ptr())
        if self.raw_rows is not None:
            b.result_rows = self.raw_rows
        self.done_flag = done_flag
        if isinstance(done_flag, int):
            b.done_flag = done_flag            # an address inside a mapped block
        elif done_flag is not None:
            if done_flag.dtype != torch.int64 or done_flag.numel() != 1:
                raise ValueError("done_flag must be a one-element int64 CUDA tensor")
            b.done_flag = done_flag.data_ptr()
        self.struct = b
        self.plan = self.engine.plan(problem, hb.feat_offset)

    def run(self, stream=None):
        """Queue frame-max + refine kernels, ordered with torch's current stream on this
        device: work queued on that stream before the call is seen by the kernels, work
        queued after it sees the results.  ``stream``: a raw hipStream_t handle to use
        instead (non-zero; 0 would mean the engine's own stream, include/ctrefine.h, which
        torch knows nothing about)."""
        torch = self.torch
        if stream:
            self.engine.refine_batch_device(self.plan, self.struct, stream)
            return
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != 0:
            self.engine.refine_batch_device(self.plan, self.struct, cur.cuda_stream)
            return
        # torch is on the legacy default stream (handle 0): the engine runs on its own stream,
        # ordered with the default stream by events on the device
        self.engine.engine_wait_stream(0)
        self.engine.refine_batch_device(self.plan, self.struct, 0)
        self.engine.stream_wait_engine(0)

    def download(self):
        """Copy the outputs back into the HostBatch arrays (synchronises)."""
        self.torch.cuda.synchronize(self.device)
        hb = self.host
        hb.params_out[...] = self.t['params_out'].cpu().numpy()
        hb.cost[...] = self.t['cost'].cpu().numpy()
        hb.status[...] = self.t['status'].cpu().numpy()
        hb.n_rounds[...] = self.t['n_rounds'].cpu().numpy()
        hb.n_iter[...] = self.t['n_iter'].cpu().numpy()
        if hb.params_std is not None:
            hb.params_std[...] = self.t['params_std'].cpu().numpy()
        return hb

    def results_tensor(self):
        """[N, n_params + 1] f64 on the device: refined parameters and the
        cost of the feature's cluster -- the rows that are gathered across ranks."""
        torch = self.torch
        n_per = torch.from_numpy(np.diff(self.host.feat_offset).astype(np.int64)).to(self.device)
        cost_rows = torch.repeat_interleave(self.t['cost'], n_per)
        return torch.cat([self.t['params_out'], cost_rows[:, None]], dim=1)

    def algorithmic_bytes(self):
        """one read of every frame + per feature 2*n_params*8
        (p0 in, params out) + per cluster 16 (cost, status, counters)."""
        hb = self.host
        return int(hb.frames.nbytes + hb.n_features * 2 * hb.params.shape[1] * 8 +
                   hb.n_clusters * 16)


def draw_frames(shape, frame_of, pos, size, max_value, n_frames=None, noise=0., seed=0,
                dtype=np.uint8, device=0, engine=None):
    """Synthetic frames drawn ON THE DEVICE (``ctr_draw_frames_device``): Gaussians by the
    rule of reference ``artificial.draw_feature`` (artificial.py:131-141: truncated to the pixel
    type, added with integer wrap-around) plus Poisson noise clipped to the pixel range
    (artificial.py:368-378).  The noise-free bytes equal ``clustertracking_amd.artificial.
    draw_gaussian``'s; the noise comes from the engine's own generator (same statistics as
    NumPy's, other bytes).

    shape: frame shape (z,) y, x; frame_of [N], pos [N, ndim], size scalar / [ndim] / [N, ndim],
    max_value scalar / [N].  Returns a torch tensor [n_frames, *shape] on the device.
    """
    import torch
    eng = engine or _lib.default_engine(device)
    dev = torch.device('cuda', device)
    ndim = len(shape)
    pos = np.ascontiguousarray(np.asarray(pos, dtype=np.float64).reshape(-1, ndim))
    n = len(pos)
    frame_of = np.array(np.broadcast_to(np.asarray(frame_of, dtype=np.int32), (n,)), order='C')
    size = np.array(np.broadcast_to(np.asarray(size, dtype=np.float64), (n, ndim)), order='C')
    max_value = np.array(np.broadcast_to(np.asarray(max_value, dtype=np.float64), (n,)), order='C')
    if n_frames is None:
        n_frames = int(frame_of.max()) + 1 if n else 0
    if n and (frame_of.min() < 0 or frame_of.max() >= n_frames):
        raise ValueError("frame index outside of the block")
    if n and (np.any(pos < 0) or np.any(pos >= np.asarray(shape))):
        raise ValueError("Position outside of image.")       # artificial.py:108-109
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.uint8), np.dtype(np.uint16)):
        raise NotImplementedError("frames are drawn as uint8 or uint16")
    if n and not np.all(max_value <= np.iinfo(dt).max):
        raise ValueError("max_value exceeds the pixel type")
    tdt = torch.uint8 if dt == np.uint8 else torch.int16
    with torch.cuda.device(dev):
        out = torch.empty((n_frames,) + tuple(shape), dtype=tdt, device=dev)
        t_fo = torch.from_numpy(frame_of).to(dev)
        t_pos = torch.from_numpy(pos).to(dev)
        t_size = torch.from_numpy(size).to(dev)
        t_mv = torch.from_numpy(max_value).to(dev)
        sy = _abi.Synth()
        sy.ndim = ndim
        sy.frame_dtype = _abi.DTYPE_CODES[dt]
        sy.n_frames = n_frames
        for a, v in enumerate(shape):
            sy.shape[a] = int(v)
        sy.n_features = n
        sy.frame_of, sy.pos, sy.size, sy.max_value = (t_fo.data_ptr(), t_pos.data_ptr(),
                                                      t_size.data_ptr(), t_mv.data_ptr())
        sy.noise = float(noise)
        sy.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        # the uploads above go through torch's stream (pageable host memory: staged copies); the
        # kernels run on the engine's own stream: a host-side synchronisation on both sides of
        # the call orders them whatever streams are current (this is a generator, not a hot path)
        torch.cuda.synchronize(dev)
        eng.draw_frames_device(sy, out.data_ptr(), 0)
        eng.synchronize()
    return out
