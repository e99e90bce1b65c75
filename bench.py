#!/usr/bin/env python3
"""Benchmarks of the ORB extraction + descriptor matching path on MI355X.  ONE JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W [--workload kitti|equirect|knn]

--workload kitti (default, the headline): one "step" = one pass of the hot path over one batch of
  synthetic frames resident in HBM: extract (pyramid, FAST + NMS, quadtree, orientation, blur, rotated
  BRIEF) of BATCH + halo frames of 1241 x 376 with 2000 keypoints / 8 levels / scale 1.2, then
  projection::match_current_and_last_frames of each of the BATCH frames against its predecessor.
  `value` = frames matched per second over all ranks (BATCH x steps x ranks / max-over-ranks time).
  Offline frame batches are independent: every rank owns a contiguous chunk of the sequence and
  re-extracts one halo frame instead of communicating ("weak" scaling, no data-path collective).
--workload equirect: the same path on 1920 x 960 frames with 4000 keypoints and the tutorial's
  rectangle masks, frame chunks sharded over the ranks the same way.
--workload knn: 2000 query descriptors against a database of 10 000 keyframes x 2000 descriptors
  (640 MB) partitioned by rows over the ranks; a step = per-shard top-2 search, ONE RCCL all-gather of
  the per-shard results (48 KB per rank) and the merge by (distance, global row) on the device.
  `value` = Hamming distances per second over the whole database ("strong" scaling).

Multi-GPU: `--gpus N` without a torch.distributed environment starts N rank processes itself (before
anything touches the GPU); under `python -m torch.distributed.run --nproc-per-node N` it uses the
ranks it is given.  torch.distributed (backend nccl = RCCL) carries the timing barrier, the maximum
over ranks and, for knn, the all-gather.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0
SHIFT_X, SHIFT_Y = 3, 1
MARGIN = 20.0

KITTI = {"name": "kitti", "width": 1241, "height": 376, "max_kp": 2000, "levels": 8, "scale": 1.2, "rects": None,
         "label": "KITTI-shaped 1241x376 mono batch, 2000 kp, 8 levels, scale 1.2, "
                  "extract + match::projection (current vs last frame, margin 20, orientation check)"}
EQUIRECT = {"name": "equirect", "width": 1920, "height": 960, "max_kp": 4000, "levels": 8, "scale": 1.2,
            "rects": [(0.0, 1.0, 0.0, 0.1), (0.0, 1.0, 0.84, 1.0), (0.0, 0.2, 0.7, 1.0), (0.8, 1.0, 0.7, 1.0)],
            "label": "equirectangular 1920x960 batch, 4000 kp, 8 levels, rectangle masks of the tutorial, "
                     "extract + match::projection (current vs last frame, margin 20, orientation check)"}


# ---- rank processes ------------------------------------------------------------------------------------
_RESULT_FD = None


def claim_stdout():
    """Keeps stdout for the one JSON line: libraries that print to file descriptor 1 (RCCL writes a
    version banner there when a communicator is created) are sent to stderr instead."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(result):
    line = (json.dumps(result) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Starts n fresh rank processes of this script (nothing in this process has touched the GPU) and
    waits for them; rank 0 inherits stdout and prints the JSON line.  A failing rank ends the others
    and makes the whole run fail."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env, stdout=out))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                for o in pending:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def rank_env(args):
    """(world, rank, local_rank) of this process; exits when --gpus contradicts the environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    return world, rank, local_rank


def planned_legs(args, world):
    """The objects rank 0's JSON line carries next to the timed value, as a function of the flags alone (the
    world size does not change them: rank 0 runs the extra legs by itself after the timed region)."""
    legs = ["roofline"]
    if args.workload == "knn":
        legs.append("parity")
        if not args.no_cpu_baseline:
            legs.append("cpu_baseline")
        return legs
    if not args.no_extras:
        legs += ["pcie_inclusive", "latency_b1"]
        if not args.no_cpu_baseline:
            legs += ["cpu_baseline", "parity"]
    return legs


# ---- extract + match workloads -------------------------------------------------------------------------
def level_pixels(cfg):
    """Pixels per pyramid level as orb_extractor.cc:267-268 sizes them (float scale chain)."""
    import numpy as np
    s = np.float32(1.0)
    out = []
    for _ in range(cfg["levels"]):
        inv = np.float32(1.0) / s
        out.append(int(round(float(np.float32(cfg["width"]) * inv))) * int(round(float(np.float32(cfg["height"]) * inv))))
        s = np.float32(s * np.float32(cfg["scale"]))
    return out


def stage_bytes(cfg):
    """Algorithmic bytes per frame, stage by stage: each stage reads its input once and writes its
    output once; patch gathers re-touch lines already counted.  A = sum of level pixels, P = level 0,
    p_last = last level."""
    px = level_pixels(cfg)
    a, p, p_last, n = sum(px), px[0], px[-1], cfg["max_kp"]
    return {
        "resize": (a - p_last) + (a - p),   # read levels 0..L-2, write levels 1..L-1
        "k_fast_cells": a,                  # read every level once
        "k_gauss7": 2 * a,                  # read + write every level
        "k_quadtree": n * 28,               # keypoints out
        "k_orient_desc": n * 32,            # descriptors out
        "match": 2 * n * 32 + n * 28,       # two descriptor sets + keypoints in
    }


STAGE_OF_KERNEL = {"k_resize_rows": "resize", "k_resize_linear": "resize", "k_batch_setup": "match", "k_grid_build": "match",
                   "k_proj_candidates": "match", "k_proj_resolve": "match"}


def make_frames_on_device(cfg, n_frames, device, seed):
    """Synthetic sequence: a window sliding by (SHIFT_X, SHIFT_Y) per frame over one seeded canvas,
    fresh noise per frame (noise drawn on the device from a seeded generator)."""
    import torch
    from openvslam_amd import synth
    w, h = cfg["width"], cfg["height"]
    canvas = synth.make_sequence(w + SHIFT_X * n_frames, h + SHIFT_Y * n_frames, 1, seed=seed, noise_sigma=0.0)[0]
    canvas = torch.from_numpy(canvas).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    frames = torch.empty((n_frames, h, w), dtype=torch.uint8, device=device)
    for t in range(n_frames):
        win = canvas[t * SHIFT_Y:t * SHIFT_Y + h, t * SHIFT_X:t * SHIFT_X + w].float()
        coarse = torch.randn((h // 2 + 1, w // 2 + 1), generator=gen, device=device) * 8.0
        noise = coarse.repeat_interleave(2, 0).repeat_interleave(2, 1)[:h, :w]
        fine = torch.randn((h, w), generator=gen, device=device) * 2.0
        frames[t] = (win + noise + fine).round().clamp(0, 255).to(torch.uint8)
    return frames


class Pipeline:
    """Extract + match of one resident batch through the C ABI (device pointers, one stream)."""

    def __init__(self, cfg, device_index, n_frames, stream=None):
        import torch
        from openvslam_amd import _abi as A
        from openvslam_amd.feature import orb_extractor, orb_params
        self.cfg = cfg
        self.dev = torch.device("cuda", device_index)
        self.n_frames = n_frames
        self.stream = stream
        self.ext = orb_extractor(orb_params(cfg["max_kp"], cfg["scale"], cfg["levels"], 20, 7, mask_rects=cfg["rects"] or ()), device=device_index)
        w, h = cfg["width"], cfg["height"]
        self.cap = self.ext.keypoint_capacity(w, h)
        self.kps = torch.zeros((n_frames, self.cap, 7), dtype=torch.float32, device=self.dev)
        self.desc = torch.zeros((n_frames, self.cap, 32), dtype=torch.uint8, device=self.dev)
        self.counts = torch.zeros((n_frames,), dtype=torch.int32, device=self.dev)
        self.reproj = torch.full((n_frames, self.cap, 2), -1e38, dtype=torch.float32, device=self.dev)
        self.matched = torch.zeros((n_frames, self.cap), dtype=torch.int32, device=self.dev)
        self.num_matches = torch.zeros((n_frames,), dtype=torch.int32, device=self.dev)
        self.shift = torch.tensor([float(SHIFT_X), float(SHIFT_Y)], dtype=torch.float32, device=self.dev)
        # zero-distortion camera: image bounds = image rectangle, 64 x 48 grid
        self.grid = A.make_grid(0.0, 0.0, float(w), float(h))
        self.sf = self.ext.get_scale_factors()
        self.device_index = device_index

    def step(self, frames):
        import torch
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._step(frames)
        else:
            self._step(frames)

    def _step(self, frames):
        import torch
        from openvslam_amd import _abi as A
        from openvslam_amd import _lib
        L = _lib.lib()
        w, h = self.cfg["width"], self.cfg["height"]
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.ext.extract_batch_dev(frames.data_ptr(), self.n_frames, w, h, w, w * h,
                                   self.kps.data_ptr(), self.desc.data_ptr(), self.counts.data_ptr(), self.cap, stream)
        # caller-side reprojection (the camera model stays outside the boundary): a point seen at
        # (x, y) in frame f-1 is expected at (x - dx, y - dy) in frame f
        torch.sub(self.kps[:-1, :, 0:2], self.shift, out=self.reproj[1:])
        _lib.check(L.ovs_match_batch_current_and_last_dev(
            self.kps.data_ptr(), self.desc.data_ptr(), self.counts.data_ptr(), self.n_frames, self.cap,
            self.reproj.data_ptr(), C.byref(self.grid), A.ptr(self.sf, A.f32p), self.cfg["levels"],
            MARGIN, 1, self.matched.data_ptr(), self.num_matches.data_ptr(), self.device_index, stream))

    def host_results(self, n):
        """(keypoints, descriptors, matched, num_matches) of the first n frames as numpy arrays."""
        import numpy as np
        from openvslam_amd import _abi as A
        cnt = self.counts[:n].cpu().numpy()
        kps = self.kps[:n].cpu().numpy().view(np.uint8).reshape(n, self.cap, 28)
        desc = self.desc[:n].cpu().numpy()
        matched = self.matched[:n].cpu().numpy()
        num = self.num_matches[:n].cpu().numpy()
        out = []
        for f in range(n):
            c = int(cnt[f])
            out.append((kps[f, :c].copy().view(A.KEYPOINT_DTYPE).reshape(-1), desc[f, :c].copy(), matched[f, :c].copy(), int(num[f])))
        return out


def _cpu_chunk(cfg, frames_host, first, stop, deadline, out, slot, keep=0, stage_out=None):
    """CPU oracle: extract + match of frames [first, stop) with frame `first` as the halo.  Keeps the
    outputs of the first `keep` frames (the parity gate)."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle
    from openvslam_amd import _abi as A
    ext = oracle.OrbExtractor(max_num_keypts=cfg["max_kp"], scale_factor=cfg["scale"], num_levels=cfg["levels"], mask_rects=cfg["rects"] or ())
    OL = oracle.lib()
    w, h = cfg["width"], cfg["height"]
    grid = A.make_grid(0.0, 0.0, float(w), float(h))
    sf = oracle.scale_table("scale_factors", cfg["levels"], cfg["scale"])
    prev = None
    done = 0
    kept = []
    match_seconds = 0.0
    extracted = 0
    for f in range(first, stop):
        kps, desc = ext.extract(frames_host[f])
        extracted += 1
        matched, num_v = None, 0
        if prev is not None:
            pk, pd = prev
            t0 = time.perf_counter()
            view = A.make_frame_view(kps, desc, sf, grid)
            reproj = np.stack([pk["x"] - SHIFT_X, pk["y"] - SHIFT_Y], axis=1).astype(np.float32)
            q = A.make_proj_queries(reproj, pk["octave"], pd, angle=pk["angle"])
            matched = np.zeros(len(kps), dtype=np.int32)
            num = C.c_uint32(0)
            oracle.check(OL.ovo_match_current_and_last_frames(view.ref(), q.ref(), None, MARGIN, A.DIR_NONE, cfg["levels"], 0.0, 1,
                                                              A.ptr(matched, A.i32p), C.byref(num)))
            match_seconds += time.perf_counter() - t0
            num_v = num.value
            done += 1
        if len(kept) < keep:
            kept.append((kps, desc, matched, num_v))
        prev = (kps, desc)
        if time.perf_counter() > deadline and done >= 2 and len(kept) >= keep:
            break
    out[slot] = done
    if stage_out is not None:
        st = ext.stage_seconds()
        stage_out.update({k: 1e3 * v / extracted for k, v in st.items()})
        stage_out["match"] = 1e3 * match_seconds / max(done, 1)
    return kept


def cpu_baseline(cfg, frames_host, seconds_budget, keep):
    """The CPU oracle (restated reference path) on a bounded sample of the same frames: first one
    thread (the reference's default build has OpenMP off), then frame-parallel over the host cores
    this process may use, contiguous chunks of at least 64 frames with a halo frame each, as the GPU
    ranks shard.  Returns the throughput figures, per-stage ms per frame of the single thread, and the
    oracle's outputs for the first `keep` frames."""
    import threading
    res = [0]
    stages = {}
    t0 = time.perf_counter()
    kept = _cpu_chunk(cfg, frames_host, 0, len(frames_host), t0 + seconds_budget / 3.0, res, 0, keep=keep, stage_out=stages)
    single = res[0] / (time.perf_counter() - t0)
    chunk = 64
    n_thr = max(1, min(16, len(os.sched_getaffinity(0)), (len(frames_host) - 1) // chunk))
    per = min((len(frames_host) - 1) // n_thr, 4 * chunk)
    res = [0] * n_thr
    t0 = time.perf_counter()
    deadline = t0 + 2.0 * seconds_budget / 3.0
    threads = [threading.Thread(target=_cpu_chunk, args=(cfg, frames_host, i * per, i * per + per + 1, deadline, res, i)) for i in range(n_thr)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    return {"fps": sum(res) / dt, "sample": sum(res), "cores": n_thr, "chunk": per, "single": single, "stages": stages}, kept


def parity_gate(first_pipe, kept):
    """Compares the device results of the first frames with the oracle's, bit for bit."""
    import numpy as np
    got = first_pipe.host_results(len(kept))
    bad = []
    for f, ((gk, gd, gm, gn), (rk, rd, rm, rn)) in enumerate(zip(got, kept)):
        if len(gk) != len(rk) or gk.tobytes() != rk.tobytes():
            bad.append("frame %d: keypoints differ" % f)
        elif not np.array_equal(gd, rd):
            bad.append("frame %d: descriptors differ" % f)
        elif rm is not None and (gn != rn or not np.array_equal(gm, rm)):
            bad.append("frame %d: matches differ" % f)
    return {"frames": len(kept), "bit_exact": not bad, "checked": "keypoints (7 fields), descriptors, matched indices, match count vs the CPU oracle",
            "mismatches": bad}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` (and of the whole path per frame) from the committed rocprofv3
    counter passes (FETCH_SIZE and WRITE_SIZE, separate runs, gfx950 corrections applied by
    tools/summarize_prof.py); the newest profiles/r*_pmc_per_kernel.csv."""
    import csv
    files = sorted((ROOT / "profiles").glob("r*_pmc_per_kernel.csv"), key=lambda p: (len(p.name), p.name))
    if not files:
        return None, None, None, None
    per_kernel, total = None, 0.0
    with open(files[-1]) as f:
        rows = list(csv.DictReader(f))
    sets = min(float(r["launches_seen"]) for r in rows)  # launch sets profiled (a kernel runs at least once per set)
    for row in rows:
        try:
            b = float(row["hbm_read_bytes(=2*FETCH_SIZE*1024)"]) + float(row["hbm_write_bytes(=WRITE_SIZE*1024)"])
        except (ValueError, KeyError):
            continue
        total += b * float(row["launches_seen"]) / sets
        if row["kernel"].split("<")[0] == kernel:
            per_kernel = b
    return per_kernel, total, files[-1].name, rows[0].get("csrc_sha256")


def stream_copy_gbs(dev):
    """Measured device-to-device copy rate (bytes read + written per second) of a 1 GiB buffer."""
    import torch
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    return 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pcie_inclusive(cfg, subs, frames, per, steps, dev, device_index):
    """The same step fed from the host.  Up: frames in page-locked memory go into one of two device input
    buffers per sub-batch on an upload stream (the upload of step s + 1 overlaps the kernels of step s).
    Down: the results of a sub-batch are packed on the device (ovs_pack_batch_results_dev: offsets = prefix
    sum of the keypoint counts, then dense keypoints / descriptors / matched indices), the offsets table
    comes down first, and once the host knows the total the payload -- not the capacity -- follows on a
    download stream of its own while the next step computes (results lag one step)."""
    import torch
    from openvslam_amd import _lib
    L = _lib.lib()
    host_frames = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True)
    host_frames.copy_(frames)
    up_stream, down_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    state = []
    nf = per + 1
    for k, sp in enumerate(subs):
        st = {"src": host_frames[k * per:k * per + nf],
              "bufs": [torch.empty((nf,) + tuple(frames.shape[1:]), dtype=torch.uint8, device=dev) for _ in range(2)],
              "uploaded": [torch.cuda.Event(), torch.cuda.Event()], "consumed": [None, None],
              "packed_ev": [torch.cuda.Event(), torch.cuda.Event()], "offsets_ev": [torch.cuda.Event(), torch.cuda.Event()],
              "downloaded": [None, None], "pending": None}
        for name, shape, dtype in (("offsets", (nf + 1,), torch.int32), ("kps", (nf * sp.cap, 7), torch.float32),
                                   ("desc", (nf * sp.cap, 32), torch.uint8), ("matched", (nf * sp.cap,), torch.int32),
                                   ("num", (nf,), torch.int32)):
            st["d_" + name] = [torch.empty(shape, dtype=dtype, device=dev) for _ in range(2)]
            st["h_" + name] = [torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(2)]
        state.append(st)
    up_bytes = sum(s["src"].numel() for s in state)
    down_bytes = [0]

    def download(st, b):
        """Payload of the step whose results sit in packed buffer b (its offsets table is on its way)."""
        st["offsets_ev"][b].synchronize()
        total = int(st["h_offsets"][b][nf])
        with torch.cuda.stream(down_stream):
            for name in ("kps", "desc", "matched"):
                st["h_" + name][b][:total].copy_(st["d_" + name][b][:total], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(down_stream)
            st["downloaded"][b] = ev
        down_bytes[0] += total * (28 + 32 + 4) + (nf + 1) * 4 + nf * 4

    def step(s_idx):
        b = s_idx & 1
        with torch.cuda.stream(up_stream):
            for st in state:
                if st["consumed"][b] is not None:
                    up_stream.wait_event(st["consumed"][b])
                st["bufs"][b].copy_(st["src"], non_blocking=True)
                st["uploaded"][b].record(up_stream)
        for st, sp in zip(state, subs):
            with torch.cuda.stream(sp.stream):
                sp.stream.wait_event(st["uploaded"][b])
                if st["downloaded"][b] is not None:
                    sp.stream.wait_event(st["downloaded"][b])  # the packed buffers of step s - 2 are on the host
                sp._step(st["bufs"][b])
                ev = torch.cuda.Event()
                ev.record(sp.stream)
                st["consumed"][b] = ev
                _lib.check(L.ovs_pack_batch_results_dev(sp.counts.data_ptr(), nf, sp.cap, sp.kps.data_ptr(), sp.desc.data_ptr(), sp.matched.data_ptr(),
                                                        st["d_offsets"][b].data_ptr(), st["d_kps"][b].data_ptr(), st["d_desc"][b].data_ptr(),
                                                        st["d_matched"][b].data_ptr(), device_index, sp.stream.cuda_stream))
                st["d_num"][b].copy_(sp.num_matches, non_blocking=True)
                st["packed_ev"][b].record(sp.stream)
            with torch.cuda.stream(down_stream):
                down_stream.wait_event(st["packed_ev"][b])
                st["h_offsets"][b].copy_(st["d_offsets"][b], non_blocking=True)
                st["h_num"][b].copy_(st["d_num"][b], non_blocking=True)
                st["offsets_ev"][b].record(down_stream)
        # the payload of the previous step, now that this step keeps the device busy
        for st in state:
            if st["pending"] is not None:
                download(st, st["pending"])
            st["pending"] = b

    def flush():
        for st in state:
            if st["pending"] is not None:
                download(st, st["pending"])
                st["pending"] = None
        torch.cuda.synchronize(dev)

    for s_idx in range(2):
        step(s_idx)
    flush()
    down_bytes[0] = 0
    t0 = time.perf_counter()
    for s_idx in range(steps):
        step(s_idx)
    flush()
    dt = (time.perf_counter() - t0) / steps
    n = per * len(subs)
    # the packed results of the last step against the capacity-strided arrays they were packed from
    ok = True
    for st, sp in zip(state, subs):
        b = (steps - 1) & 1
        off = st["h_offsets"][b].numpy()
        cnt = sp.counts.cpu().numpy()
        ok = ok and bool((off[1:] - off[:-1] == cnt).all())
        f = nf - 1
        ok = ok and bool(torch.equal(st["h_desc"][b][off[f]:off[f + 1]], sp.desc[f, :cnt[f]].cpu()))
        ok = ok and bool(torch.equal(st["h_matched"][b][off[f]:off[f + 1]], sp.matched[f, :cnt[f]].cpu()))
        ok = ok and bool(torch.equal(st["h_kps"][b][off[f]:off[f + 1]].view(torch.int32), sp.kps[f, :cnt[f]].cpu().view(torch.int32)))
    return {"value": n / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt, "h2d_gbs": up_bytes / dt / 1e9, "d2h_gbs": down_bytes[0] / steps / dt / 1e9,
            "h2d_bytes_per_frame": up_bytes / n, "d2h_bytes_per_frame": down_bytes[0] / steps / n, "packed_results_equal_strided": ok,
            "note": "frames from page-locked host memory (double-buffered uploads on an upload stream); results packed on the device "
                    "(dense keypoints, descriptors, matched indices + offsets) and brought down on a download stream while the next step runs"}


def latency_b1(cfg, frames_host):
    """One frame at a time through the host-buffer entry points (image up, keypoints and descriptors
    down, synchronous), then projection::match_current_and_last_frames against the previous frame:
    the way the reference's tracking thread calls the path."""
    import numpy as np
    from openvslam_amd import _abi as A
    from openvslam_amd import match
    from openvslam_amd.feature import orb_extractor, orb_params
    w, h = cfg["width"], cfg["height"]
    ext = orb_extractor(orb_params(cfg["max_kp"], cfg["scale"], cfg["levels"], 20, 7, mask_rects=cfg["rects"] or ()))
    matcher = match.projection(0.9, True)
    grid = A.make_grid(0.0, 0.0, float(w), float(h))
    sf = ext.get_scale_factors()
    n = min(len(frames_host), 24)
    t_ext, t_match, prev, timed = 0.0, 0.0, None, 0
    for rep in range(2):          # the first pass warms the handle up
        prev = None
        for f in range(n):
            t0 = time.perf_counter()
            k, d = ext.extract(frames_host[f])
            t1 = time.perf_counter()
            if prev is not None:
                pk, pd = prev
                view = A.make_frame_view(k, d, sf, grid)
                xy = np.stack([pk["x"] - SHIFT_X, pk["y"] - SHIFT_Y], axis=1).astype(np.float32)
                q = A.make_proj_queries(xy, pk["octave"], pd, angle=pk["angle"])
                matcher.match_current_and_last_frames(view, q, MARGIN)
                t2 = time.perf_counter()
                if rep == 1:
                    t_ext += t1 - t0
                    t_match += t2 - t1
                    timed += 1
            prev = (k, d)
    return {"extract_ms": 1e3 * t_ext / timed, "match_ms": 1e3 * t_match / timed, "frame_ms": 1e3 * (t_ext + t_match) / timed, "frames": timed,
            "note": "B = 1, host buffers in and out, synchronous (PCIe inclusive)"}


def run_extract_match(args, cfg):
    import torch
    from openvslam_amd import _lib
    world, rank, local_rank = rank_env(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from openvslam_amd.dist import shard_frames
    # the job's sequence has world x batch matched frames; this rank matches the contiguous chunk
    # [first, first + count) and extracts one more frame in front of it (the halo)
    first, count = shard_frames(args.batch * world, world, rank)
    assert count == args.batch
    n_frames = args.batch + 1
    frames = make_frames_on_device(cfg, n_frames, dev, seed=0xC0FFEE + rank)
    # The batch as `streams` contiguous sub-batches, each with its own extractor handle, buffers and
    # stream; sub-batch k starts with the last frame of sub-batch k - 1 as its halo.  The defaults make
    # every launch work on 255 + 1 = 256 frames: the kernels with one workgroup per (frame, level) or
    # per frame then fill the 256 compute units without a ragged last round, and the other streams keep
    # the device busy while such a kernel of one stream runs.
    per = args.batch // args.streams
    if per * args.streams != args.batch:
        raise SystemExit("--batch must be a multiple of --streams")
    subs = [Pipeline(cfg, local_rank, per + 1, torch.cuda.Stream(dev)) for _ in range(args.streams)]

    def step_all():
        for k, sp in enumerate(subs):
            sp.step(frames[k * per:k * per + per + 1])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step_all()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_all()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.destroy_process_group()  # everything below is rank 0's own business

    if rank != 0:
        return 0
    counts = torch.cat([sp.counts[1:] for sp in subs]).cpu().numpy()
    nm = torch.cat([sp.num_matches[1:] for sp in subs]).cpu().numpy()
    value = args.batch * args.steps * world / elapsed
    sb = stage_bytes(cfg)
    path_bytes = sum(sb.values())
    out = {
        "metric": "ORB extract+match frames/sec @KITTI 1241x376 2k kp" if cfg["name"] == "kitti"
                  else "ORB extract+match frames/sec @equirectangular 1920x960 4k kp, rectangle masks",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": cfg["label"],
                   "frames_per_step_per_gpu": args.batch, "halo_frames_per_step": args.streams, "streams": args.streams,
                   "frames_per_launch": per + 1, "sharding": "contiguous frame chunks per rank, one re-extracted halo frame, no collective",
                   "mean_keypoints_per_frame": float(counts.mean()), "mean_matches_per_frame": float(nm.mean())},
    }

    # per-kernel device time of the same step (separate, untimed pass with event pairs around every
    # launch on the launch stream; ONE sub-batch alone on ONE stream, so that no other kernel shares
    # the device while a launch is being timed)
    L = _lib.lib()
    pipe = Pipeline(cfg, local_rank, per + 1)
    pipe.step(frames[:per + 1])
    torch.cuda.synchronize(dev)
    L.ovs_timing_reset()
    L.ovs_timing_enable(1)
    prof_steps = 3
    for _ in range(prof_steps):
        pipe.step(frames[:per + 1])
    torch.cuda.synchronize(dev)
    L.ovs_timing_enable(0)
    rep = _lib.timing_report()
    per_step = {k: v[0] / prof_steps for k, v in rep.items()}
    launches = {k: v[1] // prof_steps for k, v in rep.items()}
    dominant = max(per_step, key=per_step.get)
    alg_bytes_per_launch = sb[STAGE_OF_KERNEL.get(dominant, dominant)] * (per + 1) / max(launches[dominant], 1)
    avg_ms = per_step[dominant] / max(launches[dominant], 1)
    achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, path_traffic, traffic_file, traffic_fp = (pmc_traffic(dominant) if cfg["name"] == "kitti" else (None, None, None, None))
    out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                       # counters come from a committed profile: stale when the library was built from other sources
                       "traffic_stale": (traffic_fp != _lib.source_fingerprint()) if traffic_file else None,
                       "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %d frames per launch)" % (traffic_file, per + 1)) if traffic_file else None,
                       "stream_copy_gbs": stream_copy_gbs(dev),
                       "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                       "path_bytes_per_frame": path_bytes,
                       "path_traffic_bytes_per_frame": (path_traffic / (per + 1)) if path_traffic else None,
                       "path_achieved_gbs": path_bytes * value / world / 1e9,
                       "path_frac": path_bytes * value / world / 1e9 / HBM_PEAK_GBS,
                       "kernel_ms_per_launch_set": per_step,
                       "timing_pass": "one sub-batch of %d frames alone on one stream" % (per + 1)}
    legs = planned_legs(args, world)
    if "pcie_inclusive" in legs:
        # Rank 0 alone, after the timed region (the other ranks have left): the host-fed rate, the latency of a
        # single frame, the CPU oracle on a bounded sample and the parity gate.  At N > 1 the CPU sample is
        # shorter, so that a scaling sweep stays within minutes.
        out["pcie_inclusive"] = pcie_inclusive(cfg, subs, frames, per, max(3, min(args.steps, 10)), dev, local_rank)
        host_frames = frames.cpu().numpy()
        out["latency_b1"] = latency_b1(cfg, host_frames)
        if "cpu_baseline" in legs:
            # device results of every sub-batch, taken after the legs above have reused the buffers
            step_all()
            torch.cuda.synchronize(dev)
            budget = args.cpu_seconds if world == 1 else min(args.cpu_seconds, 6.0)
            base, kept = cpu_baseline(cfg, host_frames, budget, keep=args.parity_frames)
            out["cpu_baseline"] = {"value": base["fps"], "unit": "frames/s", "cores": base["cores"], "kind": "port",
                                   "single_thread_value": base["single"],
                                   "single_thread_ms_per_frame": {k: round(v, 3) for k, v in base["stages"].items()},
                                   "sample": "%d frames of the same synthetic sequence in %d contiguous chunks of %d (one halo frame each), "
                                             "extract + match, one oracle thread per chunk" % (base["sample"], base["cores"], base["chunk"])}
            # the first frames of EVERY sub-batch (each runs on its own stream and handle) against the oracle
            gates = [parity_gate(subs[0], kept)]
            for k in range(1, len(subs)):
                res = [0]
                kept_k = _cpu_chunk(cfg, host_frames, k * per, k * per + 2, time.perf_counter() + 60.0, res, 0, keep=2)
                gates.append(parity_gate(subs[k], kept_k))
            out["parity"] = {"frames": sum(g["frames"] for g in gates), "sub_batches": len(gates), "bit_exact": all(g["bit_exact"] for g in gates),
                             "checked": gates[0]["checked"] + "; the first frames of every sub-batch (stream)",
                             "mismatches": ["sub-batch %d %s" % (k, m) for k, g in enumerate(gates) for m in g["mismatches"]]}
    emit(out)
    if "parity" in out and not out["parity"]["bit_exact"]:
        sys.stderr.write("bench.py: PARITY FAILURE: %s\n" % "; ".join(out["parity"]["mismatches"]))
        return 3
    return 0


# ---- database search workload --------------------------------------------------------------------------
KNN_KEYFRAMES, KNN_KP, KNN_QUERIES, KNN_PLANTED = 10000, 2000, 2000, 500


def knn_queries(seed=5):
    """Queries (identical on every rank) and the planted near-duplicates: query i < KNN_PLANTED is
    database row target[i] with flips[i] <= 30 bits flipped."""
    import numpy as np
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, (KNN_QUERIES, 32), dtype=np.uint8)
    target = rng.choice(KNN_KEYFRAMES * KNN_KP, KNN_PLANTED, replace=False).astype(np.int64)
    flips = rng.integers(0, 31, KNN_PLANTED)
    planted = q[:KNN_PLANTED].copy()
    for i in range(KNN_PLANTED):
        for b in rng.choice(256, flips[i], replace=False):
            planted[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return q, target, flips, planted


def run_knn(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from openvslam_amd import _abi as A
    from openvslam_amd import _lib
    from openvslam_amd.dist import shard_rows
    world, rank, local_rank = rank_env(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = str(free_port())
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # a world of one included

    total_rows = KNN_KEYFRAMES * KNN_KP
    first, n_rows = shard_rows(total_rows, world, rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + first)
    db = torch.randint(0, 256, (n_rows, 32), dtype=torch.uint8, device=dev, generator=gen)
    q_np, target, flips, planted = knn_queries()
    mine = (target >= first) & (target < first + n_rows)
    if mine.any():
        db[torch.from_numpy(target[mine] - first).to(dev)] = torch.from_numpy(planted[mine]).to(dev)
    q = torch.from_numpy(q_np).to(dev)
    nq = KNN_QUERIES
    rec = np.dtype(A.KNN2_DTYPE).itemsize
    local = torch.zeros(nq * rec, dtype=torch.uint8, device=dev)
    gathered = torch.zeros(world * nq * rec, dtype=torch.uint8, device=dev)
    merged = torch.zeros(nq * rec, dtype=torch.uint8, device=dev)
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.ovs_desc_db_create(local_rank, C.byref(h)))
    _lib.check(L.ovs_desc_db_attach_dev(h, db.data_ptr(), n_rows, first))

    def step():
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.ovs_desc_db_search_dev(h, q.data_ptr(), nq, local.data_ptr(), st))
        dist.all_gather_into_tensor(gathered, local)
        _lib.check(L.ovs_knn2_merge_dev(gathered.data_ptr(), world, nq, merged.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))

    for _ in range(args.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    dist.barrier()
    torch.cuda.synchronize(dev)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # every rank holds the merged result; it must be the same everywhere and name the planted rows
    res = merged.cpu().numpy().view(A.KNN2_DTYPE)
    check = torch.from_numpy(res.view(np.uint8).copy()).to(dev)
    ref = check.clone()
    dist.broadcast(ref, src=0)
    same_everywhere = torch.tensor([int(torch.equal(ref, check))], device=dev)
    dist.all_reduce(same_everywhere, op=dist.ReduceOp.MIN)
    planted_ok = bool(np.array_equal(res["row"][:KNN_PLANTED, 0], target) and np.array_equal(res["dist"][:KNN_PLANTED, 0], flips))

    # per-kernel time of the local search (event pairs around every launch): back to back like the timed steps,
    # after launches that bring the clocks back up from the host-side checks above
    for _ in range(2):
        _lib.check(L.ovs_desc_db_search_dev(h, q.data_ptr(), nq, local.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    L.ovs_timing_reset()
    L.ovs_timing_enable(1)
    for _ in range(max(args.steps, 3)):
        _lib.check(L.ovs_desc_db_search_dev(h, q.data_ptr(), nq, local.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    L.ovs_timing_enable(0)
    rep = {k: v[0] / max(v[1], 1) for k, v in _lib.timing_report().items()}
    rc = 0
    if rank == 0:
        search_ms = rep.get("k_knn2_mfma", rep.get("k_knn2_partial", 0.0))
        out = {"metric": "brute-force Hamming top-2 search, 2000 queries x 10k-keyframe database (20 M descriptors), distances/s",
               "value": nq * total_rows * args.steps / elapsed, "unit": "distances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u8 bits (i8 products, i32 accumulation)", "data": "synthetic",
               "config": {"workload": "kNN relocalisation: 2000 queries (25 % planted near-duplicates) vs 10 000 keyframes x 2000 descriptors, "
                                      "database rows sharded over the ranks, RCCL all-gather of per-shard top-2, device merge",
                          "rows_per_rank": n_rows, "collective": "all_gather_into_tensor, %d B per rank" % (nq * rec), "rccl_world_size": dist.get_world_size()},
               "roofline": {"bound": "mfma", "kernel": "k_knn2_mfma", "avg_launch_ms": search_ms,
                            "achieved": 2.0 * 256 * nq * n_rows / (search_ms * 1e-3) / 1e12 if search_ms else None,
                            "peak": 5000.0, "unit": "TOP/s (i8 multiply-add = 2 ops, dense peak of v_mfma_i32_16x16x64_i8)",
                            "frac": (2.0 * 256 * nq * n_rows / (search_ms * 1e-3) / 1e12 / 5000.0) if search_ms else None,
                            "traffic": None, "database_gbs": n_rows * 32 / (search_ms * 1e-3) / 1e9 if search_ms else None,
                            "kernel_ms": rep},
               "parity": {"planted_neighbours_found": planted_ok, "same_result_on_every_rank": bool(same_everywhere.item())}}
        if "cpu_baseline" in planned_legs(args, world):
            # CPU oracle on a bounded sample: the first queries against the first rows of this rank's shard
            sys.path.insert(0, str(ROOT / "tests"))
            import oracle
            sample_q, sample_rows = 16, min(4_000_000 if world == 1 else 1_500_000, n_rows)
            db_host = db[:sample_rows].cpu().numpy()
            ref_out = np.zeros(sample_q, dtype=A.KNN2_DTYPE)
            t0 = time.perf_counter()
            oracle.check(oracle.lib().ovo_knn2_search(A.ptr(db_host, A.u8p), sample_rows, 0, A.ptr(q_np, A.u8p), sample_q, A.ptr(ref_out, A.knn2p)))
            cpu_dt = time.perf_counter() - t0
            # the same sample on the device
            hs = C.c_void_p()
            _lib.check(L.ovs_desc_db_create(local_rank, C.byref(hs)))
            _lib.check(L.ovs_desc_db_attach_dev(hs, db.data_ptr(), sample_rows, 0))
            got = np.zeros(sample_q, dtype=A.KNN2_DTYPE)
            _lib.check(L.ovs_desc_db_search(hs, A.ptr(q_np, A.u8p), sample_q, A.ptr(got, A.knn2p)))
            L.ovs_desc_db_destroy(hs)
            out["parity"]["oracle_sample_bit_exact"] = bool(got.tobytes() == ref_out.tobytes())
            out["parity"]["oracle_sample"] = "%d queries x %d rows" % (sample_q, sample_rows)
            out["cpu_baseline"] = {"value": sample_q * sample_rows / cpu_dt, "unit": "distances/s", "cores": 1, "kind": "port",
                                   "sample": "%d queries x %d database rows, one oracle thread" % (sample_q, sample_rows)}
        emit(out)
        ok = planted_ok and bool(same_everywhere.item()) and out["parity"].get("oracle_sample_bit_exact", True)
        if not ok:
            sys.stderr.write("bench.py: PARITY FAILURE in the database search\n")
            rc = 3
    L.ovs_desc_db_destroy(h)
    dist.destroy_process_group()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["kitti", "equirect", "knn"], default="kitti")
    ap.add_argument("--batch", type=int, default=None, help="frames matched per step and GPU (a multiple of --streams); default 1020")
    ap.add_argument("--streams", type=int, default=4, help="contiguous sub-batches of the step's batch in flight on separate HIP streams (each re-extracts one halo frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive, B = 1 latency, CPU baseline and parity legs")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--parity-frames", type=int, default=3)
    ap.add_argument("--dry-run", action="store_true", help="print this rank's torch.distributed environment and exit (no GPU use)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # fresh rank processes, started before this process has made any GPU call
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run:
        world, rank, local_rank = rank_env(args)
        sys.stderr.write(json.dumps({"rank": rank, "local_rank": local_rank, "world": world,
                                     "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")),
                                     "legs": planned_legs(args, world) if rank == 0 else []}) + "\n")
        return 0
    claim_stdout()
    if args.batch is None:
        args.batch = 1020
    if args.workload == "knn":
        return run_knn(args)
    return run_extract_match(args, KITTI if args.workload == "kitti" else EQUIRECT)


if __name__ == "__main__":
    sys.exit(main())
