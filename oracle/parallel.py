"""Oracle on all host cores: the comp frames of a burst are independent until their contributions are summed
(reference super_resolution.py:133-173; merge.py:432-434), so one worker process per frame computes
(flow, r, kernels, that frame's num/den contribution) with the single-frame functions of this package and the parent
adds the contributions in frame order — the same float32 additions, in the same order, as the sequential
``oracle.main`` (bit-identical output; checked in tests/test_oracle_kat.py).

Test infrastructure (the ``cpu_baseline`` leg of bench.py and the full-size parity tests); never product code."""
import multiprocessing as mp
import os

import numpy as np

from .grey import compute_grey_images
from .align import init_alignment, align
from .robustness import init_robustness, compute_robustness
from .kernels import estimate_kernels
from .merge import merge as _merge, merge_ref as _merge_ref, divide

F32 = np.float32
_state = {}


def _init(ref, comp_imgs, config, fast=False, flows=None, rob=None):
    # one NumPy / BLAS thread per worker: the parallelism is over frames
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    try:
        import torch

        torch.set_num_threads(1)
    except Exception:
        pass
    cfa = np.array(config.exif.cfa_pattern)
    wb = np.array(config.exif.white_balance, dtype=np.float64)
    grey_ref = compute_grey_images(ref, config.grey_method) if config.mode == "bayer" else ref
    _state.update(ref=ref, comp=comp_imgs, config=config, cfa=cfa, wb=wb, fast=fast, flows=flows, rob_maps=rob,
                  align=init_alignment(grey_ref, config) if flows is None else None,
                  rob=init_robustness(ref, cfa, wb, config),
                  curves=(np.array(config.noise_model.std_curve, np.float64),
                          np.array(config.noise_model.diff_curve, np.float64)))


def _frame(n):
    s = _state
    cfg, img = s["config"], s["comp"][n]
    if s["flows"] is None:
        grey = compute_grey_images(img, cfg.grey_method) if cfg.mode == "bayer" else img
        flow = align(*s["align"], grey, cfg)
    else:
        flow = np.asarray(s["flows"][n], dtype=F32)
    r = (np.asarray(s["rob_maps"][n], dtype=F32) if s["rob_maps"] is not None else
         compute_robustness(img, *s["rob"], flow, s["cfa"], s["wb"], s["curves"], cfg))
    covs = estimate_kernels(img, cfg)
    H, W = img.shape
    osz = (round(cfg.scale * H), round(cfg.scale * W))
    num = np.zeros((*osz, 3), F32)
    den = np.zeros((*osz, 3), F32)
    _merge_fn(s["fast"])[0](img, flow, covs, r, num, den, s["cfa"], cfg)
    return n, flow, r, num, den


def _merge_fn(fast):
    if fast:
        from . import cfast

        return cfast.merge, cfast.merge_ref
    return _merge, _merge_ref


def main_parallel(ref_img, comp_imgs, config, workers=None, capture=None, fast=False, flows=None, rob=None):
    """Same result as ``oracle.main(ref_img, comp_imgs, config)`` (bit for bit), computed by ``workers`` processes
    (default: all host cores, at most one per comp frame).  Returns (output, debug_dict, workers_used).
    ``fast`` / ``flows`` / ``rob``: as in ``oracle.main`` (C accumulation; given flow fields instead of the alignment; given
    robustness maps instead of computing them)."""
    ref = np.asarray(ref_img, dtype=F32)
    comp_imgs = np.asarray(comp_imgs, dtype=F32)
    n = comp_imgs.shape[0]
    workers = max(1, min(workers or os.cpu_count() or 1, max(n, 1)))
    accumulate_r = bool(config.accumulated_robustness_denoiser.enabled or config.robustness.save_mask)
    H, W = ref.shape
    osz = (round(config.scale * H), round(config.scale * W))
    num = np.zeros((*osz, 3), F32)
    den = np.zeros((*osz, 3), F32)
    acc_r = np.zeros((H, W), np.float64) if accumulate_r else None
    debug = {"robustness": [], "flow": []}
    if capture is not None:
        capture.update(flow=[None] * n, r=[None] * n)
    results = {}
    if n:
        if workers == 1:
            _init(ref, comp_imgs, config, fast, flows, rob)
            it = map(_frame, range(n))
        else:
            ctx = mp.get_context("fork")  # workers inherit the burst; nothing is pickled on the way in
            pool = ctx.Pool(workers, initializer=_init, initargs=(ref, comp_imgs, config, fast, flows, rob))
            it = pool.imap_unordered(_frame, range(n))
        for k, flow, r, nk, dk in it:
            results[k] = (flow, r, nk, dk)
            while len(results) and min(results) == len(debug["flow"]):  # consume in frame order
                i = min(results)
                flow_i, r_i, nk_i, dk_i = results.pop(i)
                num += nk_i
                den += dk_i
                if accumulate_r:
                    acc_r += r_i
                debug["flow"].append(flow_i)
                debug["robustness"].append(r_i)
                if capture is not None:
                    capture["flow"][i], capture["r"][i] = flow_i, r_i
        if workers > 1:
            pool.close()
            pool.join()
        assert not results and len(debug["flow"]) == n
    cfa = np.array(config.exif.cfa_pattern)
    covs = estimate_kernels(ref, config)
    _merge_fn(fast)[1](ref, covs, num, den, cfa, config, acc_r if accumulate_r else None)
    if capture is not None:
        capture["den"] = den.copy()
    divide(num, den)
    if not config.debug:
        debug = {"robustness": [], "flow": []}
    if accumulate_r:
        debug["accumulated robustness"] = acc_r
    return num, debug, workers


def available_cores():
    """CPU cores this process may actually use: the logical CPUs of its affinity mask, capped by the cgroup CPU quota
    (containers: `cpu.max` of cgroup v2 / `cpu.cfs_quota_us` of v1) — the MI355X boxes of this project show 256 logical
    CPUs and a quota of 16: more worker processes than that only get throttled (measured: 19 / 38 / 57 / 114 processes
    give 0.43 / 0.48 / 0.38 / 0.26 Mpix/s)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _timed_crop(ref, comp, config, workers, conn):
    import time

    t0 = time.perf_counter()
    main_parallel(ref, comp, config, workers=workers)
    conn.send(time.perf_counter() - t0)
    conn.close()


def throughput_all_cores(crops, config, cores=None, capture=None):
    """The oracle on ALL host cores: `crops` = list of (ref, comp) of one shape, processed CONCURRENTLY — crop 0 in this
    process (its result is returned for the parity check), every other crop in a forked process of its own; each of them
    runs main_parallel with one worker per comp frame.  The crops are independent bursts for the timing's purpose (a
    production CPU deployment would serve independent bursts side by side the same way); frames x row slabs inside one
    burst would leave the per-frame alignment — which needs whole frames — on n_frames cores.
    Returns (output of crop 0, its debug dict, wall seconds until the LAST crop finished, worker processes used)."""
    import time

    n = max(1, len(crops[0][1]))
    cores = cores or available_cores()
    k = max(1, min(len(crops), cores // n))
    n = min(n, cores)
    ctx = mp.get_context("fork")
    procs = []
    t0 = time.perf_counter()
    for ref, comp in crops[1:k]:
        parent, child = ctx.Pipe(duplex=False)
        p = ctx.Process(target=_timed_crop, args=(ref, comp, config, n, child))  # (non-daemonic: it forks its own pool)
        p.start()
        child.close()
        procs.append((p, parent))
    out, dbg, used = main_parallel(crops[0][0], crops[0][1], config, workers=n, capture=capture)
    for p, conn in procs:
        conn.recv()
        p.join()
    return out, dbg, time.perf_counter() - t0, used * k
