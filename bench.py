"""Benchmark of the refine hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[1] -- 256 synthetic 512x512
uint8 frames PER GPU, ~200 Gaussians per frame (size 3 = radius of gyration,
diameter 13, signal 100, Poisson noise 10, seed = global frame index), isotropic
Gaussian model, default param modes.  A "step" is one pass of the hot path over
that batch with every input already resident in HBM: per-frame maximum kernel +
refine kernels (windows, masks, LM fits, re-window rounds); with N > 1 each rank
owns a contiguous block of frames (no data-path collective) and the step ends
with the RCCL gather of the result rows to rank 0.  value = cluster-fits of all
ranks / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--frames', type=int, default=256, help='frames per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='cfg2', choices=['cfg2', 'cfg3', 'cfg5'],
                    help="cfg2 is the contract's workload; cfg3 (3D stacks, --frames = stacks) and "
                         "cfg5 (dense clusters + constrained dimers) are recorded in DESIGN.md")
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                         "the N > 1 code path on a one-GPU box)")
    ap.add_argument('--size-var', action='store_true',
                    help="cfg3 with param_mode=dict(size='var') (SURVEY 8d names both): 7 variables "
                         "per feature instead of 4")
    ap.add_argument('--features', type=int, default=None,
                    help="features per frame/stack (default: the workload's own)")
    ap.add_argument('--in-flight', type=int, default=10,
                    help="batches in flight per GPU: step k starts while the slowest clusters of "
                         "steps k-1.. are still being fitted (one engine handle and one set of "
                         "output buffers per batch in flight)")
    ap.add_argument('--rehearse-collectives', action='store_true',
                    help="run the N > 1 code path (process group, staging, asynchronous gather) with one rank")
    ap.add_argument('--shard', type=int, default=None,
                    help="rehearsal on one GPU: take the frames rank SHARD of a multi-GPU run would get")
    ap.add_argument('--device-frames', action='store_true',
                    help="cfg2: draw the frames ON THE DEVICE from the seeds' truth positions "
                         "(ctr_draw_frames_device; the Poisson noise is then the engine's own generator, "
                         "not NumPy's: same statistics, other bytes)")
    ap.add_argument('--gather', default='step', choices=['step', 'final'],
                    help="N > 1: the result rows of EVERY step reach rank 0 (default) or only those of the "
                         "last step")
    ap.add_argument('--transport', default='ipc', choices=['ipc', 'rccl'],
                    help="--gather step: 'ipc' (default) = every engine writes its rows straight into its "
                         "part of rank 0's IPC-mapped inbox (peer stores over xGMI from the engine's last "
                         "kernel, ctr_batch.result_rows / done_flag: no collective per step; falls back to "
                         "'rccl' when the inbox cannot be mapped); 'rccl' = an asynchronous gather per step, "
                         "issued from the host as steps finish (ctr_query_done)")
    ap.add_argument('--layout', default='auto', choices=['auto', 'default', 'tail'],
                    help="stream layout of the engines: 'tail' = CTR_FLAG_ISOLATE_TAIL (the kernel of the likely "
                         "slow fits on a stream whose hardware queue no main stream shares); 'auto' (default) "
                         "times a few steps of both before the warm-up and keeps 'tail' where it is at least "
                         "7 %% faster")
    ap.add_argument('--single-device', action='store_true',
                    help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


sys.path.insert(0, os.path.join(ROOT, 'oracle'))


from clustertracking_amd.parallel import Inbox, open_inbox   # noqa: E402  (the inbox is product code)


def n_vars_of(problem, n):
    """optimiser variables of a cluster of n features (fitfunc.py:207-263, groups=None)"""
    from clustertracking_amd import _abi
    nv = 0
    for k in range(problem.n_params):
        m = problem.modes[k]
        nv += n if m == _abi.MODE_VAR else (1 if m == _abi.MODE_CLUSTER else 0)
    return nv


def cpu_baseline(problem, host_batch):
    """The C oracle (oracle/ctr_oracle.c: same algorithm as the engine, scalar
    C + OpenMP over clusters) timed on the host cores of this box."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ctr_oracle
    ctr_oracle.load()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    out = {}
    for key, threads in (('all', cores), ('one', 1)):
        t0 = time.perf_counter()
        ctr_oracle.run_batch(problem, host_batch, threads)
        dt = time.perf_counter() - t0
        out[key] = (host_batch.n_clusters / dt, threads, dt)
    return out


def algorithmic_flops(problem, hb, n_iter):
    """algorithmic flops of the solver iterations of one pass over the batch,
    with the MEASURED iteration counts.  Per iteration of a cluster of n features (d = ndim,
    p = mask pixels of a feature, P = union pixels, taken as n p: overlaps make it smaller):
      flops_it = sum_i p (6 d + 8) + P nv (nv + 1) + 2 P nv + 2 P + nv^3 / 3,  exps_it = sum_i p.
    Clusters of the large-cluster path (> 64 features or > 127 variables) have a BLOCK-SPARSE
    normal matrix (two features couple only where their masks overlap) and an iterative solve;
    their count is sparse too, and a lower bound:
      flops_it = n p (6 d + 8)                                    model and Jacobian rows
               + n p (npf (npf + 1) + 2 npf + 2 npf ns + 2)       diagonal blocks, J^T r, shared rows, cost
               + sum over overlapping pairs of p_ij 2 npf^2       the pair blocks (p_ij = pixels in both masks,
                                                                  from the overlap volume of the two ellipsoids
                                                                  at the start positions)
               + 2 nnz                                            ONE product with the matrix (the solve is
                                                                  conjugate gradients: ~100 products, not counted)
    Returns (flops, exps, {cluster size: (clusters, mean iterations)})."""
    d = int(problem.ndim)
    r = [int(problem.radius[a]) for a in range(d)]
    grids = np.meshgrid(*[np.arange(-x, x + 1) / float(x) for x in r], indexing='ij')
    p = int((sum(g ** 2 for g in grids) <= 1.).sum())       # refine.py:43-44, centred mask
    n = np.diff(hb.feat_offset).astype(np.int64)
    modes = [int(problem.modes[k]) for k in range(int(problem.n_params))]
    npf = sum(1 for m in modes if m == 1)
    ns = sum(1 for m in modes if m not in (0, 1))
    nv = ns + n * npf
    P = n * p
    flops_it = (n * p * (6 * d + 8) + P * nv * (nv + 1) + 2 * P * nv + 2 * P + nv ** 3 / 3.).astype(np.float64)
    large = (n > 64) | (nv > 127)
    if large.any():
        from scipy.spatial import cKDTree
        for c in np.flatnonzero(large):
            sl = slice(hb.feat_offset[c], hb.feat_offset[c + 1])
            scaled = hb.params[sl, 2:2 + d] / (2. * np.asarray(r, dtype=np.float64))
            pairs = cKDTree(scaled).query_pairs(1.0, output_type='ndarray')
            delta = 2. * np.sqrt(((scaled[pairs[:, 0]] - scaled[pairs[:, 1]]) ** 2).sum(1)) if len(pairs) else np.zeros(0)
            # common volume of two equal balls at centre distance delta (in radii), as a fraction
            # of one ball (3D) / of one disc (2D)
            if d == 3:
                frac = 1. - 0.75 * delta + delta ** 3 / 16.
            else:
                frac = (2. * np.arccos(delta / 2.) - 0.5 * delta * np.sqrt(4. - delta ** 2)) / np.pi
            p_pairs = float((p * np.clip(frac, 0., 1.)).sum())
            nnz = n[c] * npf * npf + 2 * len(pairs) * npf * npf + 2 * n[c] * npf * ns + ns * ns
            flops_it[c] = (n[c] * p * (6 * d + 8) + n[c] * p * (npf * (npf + 1) + 2 * npf + 2 * npf * ns + 2) +
                           p_pairs * 2 * npf * npf + 2 * nnz)
    it = n_iter.astype(np.float64)
    by_size = {}
    for size in np.unique(n):
        sel = n == size
        by_size[int(size)] = (int(sel.sum()), float(it[sel].mean()))
    return float((flops_it * it).sum()), float((n * p * it).sum()), by_size


def main():
    args = parse()
    # Every size class of a batch is one kernel, on one of the 4 streams of its engine
    # handle, and --in-flight batches are in flight: that needs more hardware queues than
    # ROCm's default of 4 per process, or kernels of different streams queue up behind each
    # other.  Read by the HIP runtime when it initialises, hence set before torch is imported.
    # An engine handle has 1 + 3 streams and HIP deals streams to the queues round robin: with
    # 2 queues per batch in flight (20 for the default 10) every queue carries two streams.  The
    # rate depends on that pattern -- eight batches in flight on 12 / 16 / 17 / 19 / 20 / 24 / 28
    # queues: 35 / 48 / 45 / 41 / 47 / 38 / 36 M fits/s; ten on 20 queues 49 M, and the shards with
    # one very slow fit 45 instead of 41 M -- so the SAME count is used with
    # and without a process group (from 24 queues on, every small kernel of the chain -- fill,
    # frame maximum, ordering -- takes ~0.1 ms instead of ~0.02 ms on this runtime).
    os.environ.setdefault('GPU_MAX_HW_QUEUES', str(min(20, max(4, 2 * max(1, args.in_flight)))))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (the inbox, RCCL)
    import torch
    import torch.distributed as dist
    import clustertracking_amd as cta
    from clustertracking_amd import workloads
    from clustertracking_amd.device import DeviceBatch

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if args.single_device:
        local_rank = 0
    multi = world > 1 or args.rehearse_collectives   # the exchange code path is active
    torch.cuda.set_device(local_rank)
    # The engines first: their streams are dealt to the hardware queues in the order of creation
    # (see GPU_MAX_HW_QUEUES above), so nothing that creates streams of its own (an RCCL
    # communicator) may come before them.
    from clustertracking_amd import _lib, _abi
    nfl = max(1, args.in_flight)
    engines = [_lib.default_engine(local_rank)] + [_lib.Engine(local_rank) for _ in range(nfl - 1)]
    # --transport ipc (default): the rows travel as peer stores, so the timed region needs no
    # collective at all; its control plane (row counts, barriers, the max over ranks) runs over
    # gloo, and RCCL is brought up AFTER the timed region, where it gathers the last step's rows
    # once more and must reproduce the inbox (rccl_cross_check).  An RCCL communicator that merely
    # exists during the timed region costs 12-18 % of the throughput on this workload (its streams
    # shift the mapping of the engines' 40 streams onto the 20 hardware queues; measured with one
    # rank: 47 -> 40 M fits/s) -- the rccl transport and --gather final pay that.
    ipc_plan = multi and args.gather == 'step' and args.transport == 'ipc'
    ctrl_backend = 'gloo' if (ipc_plan or args.backend == 'gloo') else 'nccl'
    coll_dev = 'cuda' if ctrl_backend == 'nccl' else 'cpu'   # where collective buffers live
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if ctrl_backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            import datetime
            dist.init_process_group('gloo', rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=600))   # (a rank that died must not hold the others for half an hour)

    # ---- this rank's shard: frames [rank*F, (rank+1)*F) of the video --------------
    shard = rank if args.shard is None else args.shard
    extra = {}
    if args.workload == 'cfg2':
        frames, f0, truth, opts = workloads.cfg2(args.frames, first_seed=shard * args.frames)
        wl_text = ("cfg2: %d frames/GPU of 512x512 uint8, 200 Gaussians/frame, size 3 (radius of "
                   "gyration), diameter 13, Poisson noise 10, isotropic Gaussian model, default "
                   "param modes" % args.frames)
    elif args.workload == 'cfg3':
        nfeat = args.features or 500
        frames, f0, truth, opts = workloads.cfg3(args.frames, first_seed=shard * args.frames,
                                                 n_features=nfeat)
        wl_text = ("cfg3: %d stacks/GPU of 64x128x128 uint8, %d Gaussians/stack, size (2,4,4), "
                   "diameter (9,17,17), anisotropic Gaussian model, %s"
                   % (args.frames, nfeat, "param_mode size='var'" if args.size_var else "default param modes"))
        if args.size_var:
            extra['param_mode'] = dict(size='var')
    else:
        frames, f0, truth, opts = workloads.cfg5(args.frames, first_seed=shard * args.frames)
        extra['constraints'] = cta.constraints.dimer(6., 2)
        wl_text = ("cfg5: %d frames/GPU of 512x512 uint8, 36 compact clusters of 2/8-16 Gaussians, "
                   "size 3, diameter 13, dimers constrained to 2*size" % args.frames)
    if args.device_frames:
        if args.workload != 'cfg2':
            raise SystemExit("--device-frames is implemented for cfg2")
        from clustertracking_amd.device import draw_frames
        t0 = time.perf_counter()
        dev_frames = draw_frames(frames.shape[1:], f0['frame'].values.astype(np.int32), truth, 3., 100.,
                                 n_frames=frames.shape[0], noise=10., seed=12345 + shard, device=local_rank)
        torch.cuda.synchronize()
        extra_info = {"frames_drawn_on_device_s": time.perf_counter() - t0}
        frames = dev_frames.cpu().numpy()       # (the host copy feeds prepare_batch and the CPU legs)
        wl_text += "; frames drawn on the device (ctr_draw_frames_device)"
    else:
        extra_info = {}
    reader = cta.ArrayReader(frames)
    t0 = time.perf_counter()
    prep = cta.prepare_batch(f0, reader, opts['diameter'], **extra)
    t_host_prep = time.perf_counter() - t0
    if nfl > 1:     # several batches in flight: machine time per cluster before one-batch latency
        prep.problem.flags |= _abi.FLAG_THROUGHPUT
    pad_rows = 0
    if multi and args.gather == 'step':
        # every rank sends the same number of rows: the largest feature count of any rank
        cnt_t = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(world)]
        dist.all_gather(cnt_t, torch.tensor([prep.batch.n_features], dtype=torch.int64, device=coll_dev))
        pad_rows = max(int(c.item()) for c in cnt_t)
    # --transport ipc: rank 0 owns inbox[world, in flight, pad_rows, n_params + 1] and
    # seq[world, in flight]; every rank maps them (torch's CUDA IPC: hipIpcOpenMemHandle) and hands
    # its engines their slice as result_rows / done_flag
    inbox = None
    use_ipc = False
    data_group, data_dev = None, coll_dev        # where the RCCL / gloo gathers of rows run
    if ipc_plan:
        use_ipc, inbox = open_inbox(torch, dist, engines[0], rank, world, nfl, pad_rows,
                                    prep.batch.params.shape[1] + 1, coll_dev)
        if not use_ipc and args.backend == 'nccl':
            data_group, data_dev = dist.new_group(backend='nccl'), 'cuda'   # the rccl transport after all
    dbs = []
    for j, e in enumerate(engines):
        if use_ipc:
            dbs.append(DeviceBatch(prep.problem, prep.batch, device=local_rank, engine=e,
                                   result_rows=(inbox.rows_addr(rank, j), inbox.pad_rows),
                                   done_flag=inbox.seq_addr(rank, j)))
        else:
            dbs.append(DeviceBatch(prep.problem, prep.batch, device=local_rank, engine=e, result_rows=pad_rows))
    db = dbs[0]
    n_fits = prep.batch.n_clusters
    n_feat = prep.batch.n_features

    # A step = one pass of the hot path over the batch.  Steps go round robin over the
    # engines; an engine's own stream keeps its steps in order, different engines overlap
    # on the GPU, so the slowest clusters of one step (a bin lasts as long as its slowest
    # cluster) no longer hold up the next steps.
    # The only exchange of the path (north_star: RCCL only for the result gather): the result
    # rows of every step go from every rank to rank 0, inside the timed region.  With batches in
    # flight a gather must not park a stream in a device-side wait for a slow step (it blocks the
    # streams sharing its hardware queue) nor stall the host behind the slowest step (round 1:
    # 15-30 % of the throughput either way).  So the HOST hands finished steps on: after every
    # launch it asks, without blocking, which of the oldest steps in flight have finished
    # (ctr_query_done), packs their rows into that slot's send buffer and issues the gather
    # asynchronously; a slot is only waited for when its turn comes again before its step has
    # been handed on.  --gather final: the rows of the last step only (round 1's protocol).
    step_no = [0]
    send, gather_buf = None, None
    pending = []          # slots whose step has been launched and not yet gathered, oldest first
    slot_work = [None] * max(1, args.in_flight)   # the asynchronous gather that last used a slot's buffers
    if multi:
        counts = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([n_feat], dtype=torch.int64, device=coll_dev))
        counts = [int(c.item()) for c in counts]
        width = prep.batch.params.shape[1] + 1
        pad = max(counts)
        send = torch.zeros((pad, width), dtype=torch.float64, device=data_dev)
        if rank == 0:
            gather_buf = [torch.empty((pad, width), dtype=torch.float64, device=data_dev)
                          for _ in range(world)]
            gather_bufs = [[torch.empty((pad, width), dtype=torch.float64, device=data_dev)
                            for _ in range(world)] for _ in range(nfl if not use_ipc else 0)]
        # cluster of every feature row, resident on the device: cost[row_cluster] = cost per row
        row_cluster = torch.from_numpy(np.repeat(np.arange(n_fits, dtype=np.int64),
                                                 np.diff(prep.batch.feat_offset))).to(db.device)

    def hand_on(slot):
        """pack the rows of the finished step of `slot` and issue their gather (asynchronous)"""
        d = dbs[slot]
        # the engine has written the rows of the result table (params | cost of the row's
        # cluster, ctr_batch.result_rows) into the slot's send block itself: nothing to pack
        buf = d.t['result_rows'] if data_dev == 'cuda' else d.t['result_rows'].cpu()
        slot_work[slot] = dist.gather(buf, gather_bufs[slot] if rank == 0 else None, dst=0, async_op=True,
                                      group=data_group)

    def poll(block_slot=None):
        """hand on the oldest finished steps; with block_slot, everything up to that slot"""
        while pending:
            slot = pending[0]
            if not dbs[slot].engine.query_done():
                if block_slot is None:
                    return
                dbs[slot].engine.synchronize()
            pending.pop(0)
            hand_on(slot)
            if slot == block_slot:
                block_slot = None

    per_step_gather = multi and args.gather == 'step' and not use_ipc

    def step():
        slot = step_no[0] % nfl
        d = dbs[slot]
        step_no[0] += 1
        if per_step_gather:
            if slot in pending:
                poll(block_slot=slot)                   # its previous step has not been handed on yet
            if slot_work[slot] is not None:
                slot_work[slot].wait()                  # the gather that reads its send block (long done)
                d.engine.engine_wait_stream(0)
        if use_ipc:
            d.struct.done_value = step_no[0]            # rank 0 sees seq[rank, slot] = number of the step
        d.engine.refine_batch_device(d.plan, d.struct, 0)   # the engine's own stream
        if per_step_gather:
            pending.append(slot)
            poll()

    def flush_gathers():
        if pending:
            poll(block_slot=pending[-1])
        for w in slot_work:
            if w is not None:
                w.wait()

    def final_gather():
        d = dbs[(step_no[0] - 1) % nfl]                 # the batch of the last step
        d.engine.stream_wait_engine(0)                  # torch's stream waits for that engine
        if data_dev == 'cuda':
            send[:n_feat, :width - 1].copy_(d.t['params_out'])
            send[:n_feat, width - 1].copy_(d.t['cost'][row_cluster])
        else:                                           # gloo (rehearsal): through host memory
            send[:n_feat, :width - 1].copy_(d.t['params_out'].cpu())
            send[:n_feat, width - 1].copy_(d.t['cost'][row_cluster].cpu())
        dist.gather(send, gather_buf, dst=0, group=data_group)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    # set-up, not a step: the first call of a handle sets kernel attributes and loads code
    for d in dbs:
        d.engine.refine_batch_device(d.plan, d.struct, 0)
    torch.cuda.synchronize()
    # Scheduling choice, made per rank before the warm-up (untimed): a batch that holds one fit
    # far slower than the rest (cfg 2: two start positions on one real feature, 300+ iterations,
    # 4 ms in the pairs kernel of every step -- 2 of the 8 shards of an 8-GPU run have one) is
    # served better with CTR_FLAG_ISOLATE_TAIL (include/ctrefine.h); which it is shows in a few
    # steps of each.  Results do not depend on the flag.
    layout = 'default'
    has_large = prep.batch.n_clusters > 0 and int(np.diff(prep.batch.feat_offset).max()) > 64   # (their plans own GBs of workspace)
    if nfl > 1 and args.layout != 'default' and not has_large:
        import copy
        prob_tail = copy.copy(prep.problem)
        prob_tail.flags |= _abi.FLAG_ISOLATE_TAIL
        plans = {'default': [d.plan for d in dbs],
                 'tail': [d.engine.plan(prob_tail, prep.batch.feat_offset) for d in dbs]}
        trial = {}
        if args.layout == 'auto':
            for name in ('default', 'tail', 'default', 'tail'):
                for d, pl in zip(dbs, plans[name]):
                    d.plan = pl
                for d in dbs:                                   # (attributes, code)
                    d.engine.refine_batch_device(d.plan, d.struct, 0)
                torch.cuda.synchronize()
                t_try = time.perf_counter()
                for k_try in range(2 * nfl):
                    d = dbs[k_try % nfl]
                    d.engine.refine_batch_device(d.plan, d.struct, 0)
                torch.cuda.synchronize()
                trial[name] = min(trial.get(name, 1e9), (time.perf_counter() - t_try) / (2 * nfl))
            layout = 'tail' if trial['tail'] < 0.93 * trial['default'] else 'default'
        else:
            layout = args.layout
        for d, pl in zip(dbs, plans[layout]):
            d.plan = pl
        layout_info = {"layout": layout, "layout_trial_ms_per_step": {k2: v * 1e3 for k2, v in trial.items()}}
    else:
        layout_info = {"layout": layout}
    for _ in range(args.warmup):
        step()
    if multi and not use_ipc:
        flush_gathers()
        final_gather()      # (the first gather sets up the point-to-point channels)
    fence()
    fm_ms, rf_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if per_step_gather:
        flush_gathers()
    elif multi and not use_ipc:
        final_gather()
    fence()                 # (--transport ipc: the rows are in rank 0's inbox once every rank's engines are done)
    elapsed = time.perf_counter() - t0
    # kernel durations of ONE batch running alone, default scheduling (no throughput flag):
    # HIP events recorded on the launch stream inside the library
    import copy
    prob_alone = copy.copy(prep.problem)
    prob_alone.flags &= ~_abi.FLAG_THROUGHPUT
    db_alone = DeviceBatch(prob_alone, prep.batch, device=local_rank, engine=engines[0]) if nfl > 1 else db
    for _ in range(5):
        db_alone.run()
        a, b = db_alone.engine.last_kernel_ms()
        fm_ms.append(a)
        rf_ms.append(b)
    fence()

    # every batch in flight must have produced the same table
    ran = min(nfl, step_no[0])
    copies_same = all(bool(torch.equal(db.t['params_out'], d.t['params_out'])) and
                      bool(torch.equal(db.t['status'], d.t['status'])) for d in dbs[1:ran])
    gather_ok = None
    if multi and rank == 0:
        # what arrived in the last gather: rank 0's own rows must be its results, every other
        # rank's rows finite positions inside its frames
        if use_ipc:
            # the inbox: every rank's slot of the last step carries that step's number, rank 0's
            # rows are its results, the others' rows finite positions
            slot_l = (step_no[0] - 1) % nfl
            last = [torch.from_numpy(inbox.read_rows(r, slot_l, counts[r])) for r in range(world)]
            # EVERY slot: it must hold the number of the last step that was mapped to it, and finite
            # rows from every rank (nothing reads or acknowledges the slots during the timed region:
            # a consumer on rank 0 polls seq; reusing a slot needs its acknowledgement, ctrefine.h)
            seq = inbox.read_seq()
            seq_ok = True
            for sl_i in range(min(nfl, step_no[0])):
                want = step_no[0] - ((step_no[0] - 1 - sl_i) % nfl)
                seq_ok = seq_ok and bool((seq[:, sl_i] == want).all())
                for r in range(world):
                    seq_ok = seq_ok and bool(np.isfinite(inbox.read_rows(r, sl_i, counts[r])[:, 2:4]).all())
        else:
            last = gather_bufs[(step_no[0] - 1) % nfl] if per_step_gather else gather_buf
            seq_ok = True
        own = last[0][:n_feat, :width - 1].to(db.device)
        gather_ok = seq_ok and bool(torch.equal(own, dbs[(step_no[0] - 1) % nfl].t['params_out']))
        for r in range(1, world):
            rows = last[r][:counts[r], 2:4]
            gather_ok = gather_ok and bool(torch.isfinite(rows).all()) and bool((rows > -20).all())

    # --transport ipc: RCCL comes up only now, outside the timed region, and gathers the rows of
    # the last step once more over xGMI: what it delivers must be what the inbox holds
    rccl_check = None
    if use_ipc and args.backend == 'nccl':
        try:
            g_rccl = dist.new_group(backend='nccl')
            d_last = dbs[(step_no[0] - 1) % nfl]
            send_c = torch.zeros((pad, width), dtype=torch.float64, device='cuda')
            send_c[:n_feat, :width - 1].copy_(d_last.t['params_out'])
            send_c[:n_feat, width - 1].copy_(d_last.t['cost'][row_cluster])
            bufs = [torch.empty_like(send_c) for _ in range(world)] if rank == 0 else None
            dist.gather(send_c, bufs, dst=0, group=g_rccl)
            torch.cuda.synchronize()
            if rank == 0:
                slot_l = (step_no[0] - 1) % nfl
                rccl_check = all(bool(np.array_equal(bufs[r][:counts[r]].cpu().numpy(),
                                                     inbox.read_rows(r, slot_l, counts[r])))
                                 for r in range(world))
        except Exception as e:   # noqa: BLE001 (reported in the bench line, the measurement stands)
            rccl_check = "failed: %r" % (e,)

    t_all = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    fits_all = torch.tensor([float(n_fits)], dtype=torch.float64, device=coll_dev)
    if multi:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(fits_all, op=dist.ReduceOp.SUM)
    elapsed = float(t_all.item())
    total_fits = float(fits_all.item())

    # ---- correctness of what was timed (rank 0): oracle on the same batch ---------
    hb = db.download()
    n_fail = int((hb.status != 0).sum())
    gpu_out, gpu_status, gpu_cost = hb.params_out.copy(), hb.status.copy(), hb.cost.copy()
    mean_iters = float(hb.n_iter.mean())
    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_fits * args.steps / elapsed
        alg_bytes = db.algorithmic_bytes()
        rf_alone = float(np.median(rf_ms)) * 1e-3
        fm = float(np.median(fm_ms)) * 1e-3
        # duration of the refine stage per launch: alone when one batch is in flight; with
        # several in flight the stages of neighbouring steps overlap and one completes per
        # step time (steady state)
        rf = rf_alone if nfl == 1 else elapsed / args.steps
        peak = 8000.0
        traffic = None
        traffic_source = None
        tf = os.path.join(ROOT, 'profiles', 'traffic_cfg2.json')
        if args.workload == 'cfg2' and args.frames == 256 and os.path.exists(tf):
            # HBM bytes of the refine kernels from the COMMITTED rocprofv3 PMC passes
            # (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction) -- counters cannot be
            # read from inside this process, so this is a profile of the same command, not of
            # this run; see the file for the build it was taken on
            tj = json.load(open(tf))
            traffic = tj['refine_kernels']['bytes_corrected']
            traffic_source = "committed profile (%s), not this run" % tj.get('profile', 'profiles/traffic_cfg2.json')
        tf3 = os.path.join(ROOT, 'profiles', 'traffic_cfg3.json')
        if args.workload == 'cfg3' and args.frames == 64 and not args.size_var and os.path.exists(tf3):
            tj = json.load(open(tf3))     # (same: the committed FETCH_SIZE / WRITE_SIZE passes of the 64-stack step)
            traffic = tj['refine_large_kernel']['bytes_corrected']
            traffic_source = "committed profile (%s), not this run" % tj.get('profile', 'profiles/traffic_cfg3.json')
        n_per_cluster = np.diff(hb.feat_offset)
        if hb.n_clusters and int(n_per_cluster.max()) > 64:
            kernel_names = ("refine stage: refine_large_kernel<%d,%s> (leader workgroup per cluster; %.0f %% of the features) + the "
                            "small / block kernels for the rest" % (prep.problem.ndim, 'iso' if prep.problem.isotropic else 'aniso',
                                                                    100. * n_per_cluster[n_per_cluster > 64].sum() / max(n_feat, 1)))
        elif args.workload == 'cfg5':
            kernel_names = "refine stage: refine_block_kernel<2,iso,NT,W> + its constrained instantiations (concurrent streams)"
        else:
            kernel_names = "refine stage: refine_small_kernel<2,1|2> + refine_block_kernel<2,iso,NT,W> (concurrent streams)"
        flops, exps, by_size = algorithmic_flops(prep.problem, hb, hb.n_iter)
        # engine vs the reference's algorithm over the WHOLE workload (all 41 033 clusters, SLSQP with
        # the default and with a converged tolerance): tools/full_parity.py on the GPU box, committed
        full_parity = None
        pf = os.path.join(ROOT, 'profiles', 'r03_parity_full_cfg2.json')
        if args.workload == 'cfg2' and args.frames == 256 and shard == 0 and os.path.exists(pf):
            pj = json.load(open(pf))
            full_parity = {"source": "committed profile profiles/r03_parity_full_cfg2.json (tools/full_parity.py), not this run"}
            for key in ('vs_reference_algorithm_default_tol_1e-6', 'vs_reference_algorithm_converged_tol_1e-14',
                        'reference_A_vs_B'):
                full_parity[key] = {k2: v2 for k2, v2 in pj[key].items() if k2 != 'clusters_above_1e-3_px'}
        fp64_peak = 78.6     # TFLOP/s, FP64 vector (MI355X_MICROARCH.md / SURVEY.md 8d)
        valu = None
        vf = os.path.join(ROOT, 'profiles', 'valu_cfg2.json')
        if args.workload == 'cfg2' and args.frames == 256 and os.path.exists(vf):
            valu = json.load(open(vf))
        result = {
            "metric": "cluster-fits/sec", "value": value, "unit": "cluster-fits/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_text,
                       "frames_per_gpu": args.frames, "cluster_fits_per_gpu": n_fits,
                       "features_per_gpu": n_feat, "parallelism": "frames sharded, %d rank(s)" % world},
            "features_per_s": value * n_feat / max(n_fits, 1),
            "failed_clusters": n_fail,
            "mean_solver_iterations": mean_iters,
            "parity_full_workload": full_parity,
            "roofline": {"bound": "hbm", "kernel": kernel_names,
                         "achieved": alg_bytes / rf / 1e9, "peak": peak, "unit": "GB/s",
                         "frac": alg_bytes / rf / 1e9 / peak, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": rf * 1e3,
                         "kernel_ms_one_batch_alone": rf_alone * 1e3},
            "roofline_frame_max": {"bound": "hbm", "kernel": "frame_max_kernel",
                                   "achieved": hb.frames.nbytes / fm / 1e9, "peak": peak,
                                   "unit": "GB/s", "frac": hb.frames.nbytes / fm / 1e9 / peak,
                                   "kernel_ms": fm * 1e3},
            # the bound that matters for this path: the LM iterations run on chip,
            # FP64 vector / transcendental bound; algorithmic flops with the measured iterations
            "roofline_fp64_valu": {"bound": "valu-fp64", "achieved": flops / rf / 1e12, "peak": fp64_peak,
                                   "unit": "TFLOP/s", "frac": flops / rf / 1e12 / fp64_peak,
                                   "algorithmic_flops_per_launch": flops, "exp_per_launch": exps,
                                   "iterations_by_cluster_size": {str(k): {"clusters": v[0], "mean_iterations": v[1]}
                                                                  for k, v in sorted(by_size.items())},
                                   "valu_busy": valu,
                                   "note": "flops_it = sum p(6d+8) + P nv(nv+1) + 2 P nv + 2P + nv^3/3 per solver "
                                           "iteration (SURVEY.md 8d), P = n p; x the measured n_iter of every cluster"},
            "host_prepare_s": t_host_prep,
            **extra_info,
            "gather_checked": gather_ok,
            "gather": (args.gather if multi else None),
            "gather_transport": (("ipc inbox on rank 0 (peer stores over xGMI, no collective per step)" if use_ipc
                                  else "%s gather per step" % ('rccl' if data_dev == 'cuda' else 'gloo'))
                                 if multi and args.gather == 'step' else None),
            "rccl_cross_check": rccl_check,
            "batches_in_flight": nfl,
            **layout_info,
            "in_flight_results_identical": copies_same,
        }
        big_clusters = int(np.diff(hb.feat_offset).max()) > 127 if hb.n_clusters else False
        if world == 1 and not args.no_cpu_baseline and big_clusters:
            # cfg 3 at its stated density: one cluster of ~500 features per stack.  Neither CPU leg
            # finishes one such fit within minutes (SLSQP on 2001 variables x 10 re-window rounds:
            # hours; the C oracle: 445 s for stack 0, measured once in the build container for
            # tests/golden/cfg3_500_oracle.npz).  Timed here, as the bounded sample: TWO solver
            # iterations of the oracle (dense Cholesky LM, one thread) on the first large cluster of
            # the batch; the rate of whole fits follows with the iterations the engine needed for it.
            import copy
            import ctr_oracle
            n_per_c = np.diff(hb.feat_offset)
            c0 = int(np.flatnonzero(n_per_c > 127)[0])
            rows0 = np.arange(hb.feat_offset[c0], hb.feat_offset[c0 + 1])
            one = _abi.HostBatch(hb.frames, hb.frame_index[c0:c0 + 1], np.array([0, len(rows0)], hb.feat_offset.dtype),
                                 hb.params[rows0], hb.low[rows0], hb.high[rows0])
            p2 = copy.copy(prep.problem)
            p2.max_iter, p2.solver_maxiter = 1, 2
            t0 = time.perf_counter()
            ctr_oracle.run_batch(p2, one, 1)
            dt2 = time.perf_counter() - t0
            it_cpu = max(int(one.n_iter[0]), 1)
            it_gpu = max(int(hb.n_iter[c0]), 1)
            result["cpu_baseline"] = {
                "value": 1. / (dt2 / it_cpu * it_gpu), "unit": "cluster-fits/s", "cores": 1, "kind": "port",
                "seconds_per_solver_iteration": dt2 / it_cpu,
                "engine_iterations_of_that_cluster": it_gpu,
                "sample": "%d solver iterations of oracle/ctr_oracle.c (the engine's bounded LM with a dense Cholesky, one "
                          "thread) on the first large cluster of the batch (%d features, %d variables): %.1f s; value = "
                          "1 / (seconds per iteration x the %d iterations the engine needed for that cluster).  A whole "
                          "fit of stack 0 took the oracle 445 s in the build container (tests/golden/make_golden_cfg3.py)"
                          % (it_cpu, len(rows0), n_vars_of(prep.problem, len(rows0)), dt2, it_gpu),
                "reference": "the reference hands such a cluster to SciPy's SLSQP with maxiter=100 (refine.py:373-377): on "
                             "2001 variables its BFGS model cannot converge within that, success=False, the whole stack "
                             "comes back NaN (observed for the 75 / 90-feature fixtures' larger siblings in the build "
                             "container; hours per stack)"}
        if world == 1:
            # the drop-in call end to end (host buffers in, DataFrame out): prepare on the host,
            # ctr_refine_batch incl. the PCIe copies of frames and tables, vectorised write-back;
            # never `value` (whose inputs are resident in HBM)
            e2e = {}
            for labels in ('device', 'reference'):
                if labels == 'reference' and (big_clusters or args.workload != 'cfg2'):
                    continue
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    pr = cta.prepare_batch(f0, reader, opts['diameter'], cluster_labels=labels,
                                           device=local_rank, **extra)
                    t1 = time.perf_counter()
                    engines[0].refine_batch(pr.problem, pr.batch)
                    t2 = time.perf_counter()
                    cta.write_back(pr)
                    t3 = time.perf_counter()
                    if best is None or t3 - t0 < best[0]:
                        best = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
                e2e[labels] = {"fits_per_s": n_fits / best[0], "prepare_s": best[1],
                               "engine_call_incl_pcie_s": best[2], "write_back_s": best[3]}
            result["drop_in_end_to_end"] = dict(e2e, note="refine_leastsq's three stages on one batch of the "
                                                "workload, best of 3; cluster_labels='reference' labels on the host with "
                                                "ids equal to the reference's, 'device' labels the same partition on the GPU")
        if world == 1 and not args.no_cpu_baseline and not big_clusters:
            pos = slice(2, 2 + frames.ndim - 1)
            # (a) the reference's own algorithm restated (NumPy objective + SciPy SLSQP,
            #     oracle/ref_numpy.py), one thread, on a bounded sample of the same workload
            import ref_numpy
            n_sample_frames = max(1, min(args.frames, 8 if args.workload == 'cfg2' else 2))
            sample = np.flatnonzero(hb.frame_index < n_sample_frames)
            t0 = time.perf_counter()
            ref_numpy.run_batch(prep.problem, hb, clusters=sample)
            dt_py = time.perf_counter() - t0
            rows = np.concatenate([np.arange(hb.feat_offset[c], hb.feat_offset[c + 1]) for c in sample])
            ok_c = (hb.status[sample] == 0) & (gpu_status[sample] == 0)
            # a handful of clusters (near-coincident features) have several local minima and
            # the two solvers may settle in different ones (either may have the lower cost):
            # they are counted, and the position statistics are over the clusters in the same one
            same_min = ok_c & (np.abs(gpu_cost[sample] - hb.cost[sample]) <= 1e-5 * np.abs(hb.cost[sample]))
            n_per = np.diff(hb.feat_offset)[sample]
            d_all = np.abs(gpu_out[rows][:, pos] - hb.params_out[rows][:, pos]).max(1)
            d = (gpu_out[rows][:, pos] - hb.params_out[rows][:, pos])[np.repeat(same_min, n_per)]
            d_unf = (gpu_out[rows][:, pos] - hb.params_out[rows][:, pos])[np.repeat(ok_c, n_per)]
            others = []
            for c_i in np.flatnonzero(ok_c & ~same_min):
                cl = int(sample[c_i])
                sl = slice(hb.feat_offset[cl], hb.feat_offset[cl + 1])
                a_pos, b_pos = gpu_out[sl][:, pos], hb.params_out[sl][:, pos]
                # the same positions with the labels permuted? (two start positions on overlapping
                # features can trade places: the masks follow the labels, so the costs differ)
                import itertools
                best_perm = min((float(np.abs(a_pos - b_pos[list(pm)]).max())
                                 for pm in itertools.permutations(range(len(a_pos)))), default=0.) \
                    if len(a_pos) <= 6 else None
                tr = truth[prep.order[sl]] if truth is not None else None
                others.append({"cluster": cl, "features": int(n_per[c_i]),
                               "cost_engine": float(gpu_cost[cl]), "cost_slsqp": float(hb.cost[cl]),
                               "lower_cost": "engine" if gpu_cost[cl] < hb.cost[cl] else "slsqp",
                               "max_dpos_px": float(np.abs(a_pos - b_pos).max()),
                               "max_dpos_px_best_label_assignment": best_perm,
                               "max_err_vs_truth_px": None if tr is None else {
                                   "engine": float(np.abs(a_pos - tr).max()), "slsqp": float(np.abs(b_pos - tr).max())}})
            result["parity_vs_scipy_slsqp_px"] = {
                "rmse": float(np.sqrt(np.mean(d ** 2))), "max": float(np.abs(d).max()),
                "rmse_unfiltered": float(np.sqrt(np.mean(d_unf ** 2))),
                "max_unfiltered": float(np.abs(d_unf).max()),
                "other_minimum_clusters": others,
                "median_all": float(np.median(d_all[np.repeat(ok_c, n_per)])),
                "clusters": int(len(sample)), "clusters_same_minimum": int(same_min.sum()),
                "clusters_other_minimum": int((ok_c & ~same_min).sum()),
                "failed_here_not_there": int(((gpu_status[sample] != 0) & (hb.status[sample] == 0)).sum()),
                "failed_there_not_here": int(((gpu_status[sample] == 0) & (hb.status[sample] != 0)).sum()),
                "note": "engine vs the reference algorithm with its default SLSQP tol=1e-6 "
                        "(north_star: <= 1e-3 px); rmse/max over clusters whose cost agrees to 1e-5, "
                        "rmse_unfiltered/max_unfiltered over all clusters both fit; every cluster that "
                        "ends in another local minimum is listed with both costs"}
            # (b) the C oracle (same LM as the engine, scalar C + OpenMP over clusters), full workload
            base = cpu_baseline(prep.problem, prep.batch)
            both = (hb.status == 0) & (gpu_status == 0)
            result["status_equal_oracle"] = bool((hb.status == gpu_status).all())
            ok_rows = np.repeat(both, np.diff(hb.feat_offset))
            dpos = (gpu_out[:, pos] - hb.params_out[:, pos])[ok_rows]
            result["parity_vs_oracle_px"] = {"rmse": float(np.sqrt(np.mean(dpos ** 2))),
                                             "max": float(np.abs(dpos).max())}
            result["cpu_baseline"] = {
                "value": len(sample) / dt_py, "unit": "cluster-fits/s", "cores": 1, "kind": "port",
                "sample": "the first %d frames of the workload (%d cluster-fits, %.1f s): "
                          "oracle/ref_numpy.py = the reference's algorithm (NumPy objective + "
                          "SciPy SLSQP tol=1e-6, Python loop per cluster)" % (
                              n_sample_frames, len(sample), dt_py),
                "reference_python_fits_per_s_survey": 109.0}
            result["cpu_baseline_native_lm"] = {
                "value": base['all'][0], "unit": "cluster-fits/s", "cores": base['all'][1],
                "kind": "port", "one_thread_value": base['one'][0],
                "sample": "the full workload of one GPU (%d cluster-fits): oracle/ctr_oracle.c = "
                          "the engine's bounded LM in scalar C, OpenMP over clusters" % n_fits}
        print(json.dumps(result), flush=True)
    if multi:
        if use_ipc:
            # importers unmap, then the owner frees
            torch.cuda.synchronize()
            if rank != 0:
                inbox.release()
            dist.barrier()
            if rank == 0:
                inbox.release()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
