#!/usr/bin/env python3
"""bench.py — whole-job throughput of the rasterize() hot path on MI355X.

Workload (BASELINE.json configs[3], the one its metric "triangles/sec ... at 4096x4096" is quoted on):
10 M synthetic random triangles, 4096x4096 RGB framebuffer + fp64 z-buffer, flat shader.
A "step" is one full frame: clear -> setup -> stable tile binning -> block raster (one wavefront per 8x8 pixels, depths in
registers) -> block-out, with the triangle stream already resident in HBM when the timed region starts.  The timed loop runs
with the library's event profiling OFF (nothing waits inside it); a second, short loop with profiling on supplies the
per-phase times and the k_raster launch duration of the roofline entry.

N > 1 (launched by torch.distributed.run, one rank per GPU): strong scaling on the SAME frame — rank r owns
framebuffer rows [r*H/N, (r+1)*H/N), every rank streams all triangles through setup (replicating 96 B/triangle of
reads is cheaper than moving records over xGMI) and the colour strips are joined with one RCCL all-gather.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(clip, col, W, H, sample):
    """Single-thread CPU rate on a bounded prefix of the same workload: the reference's own rasterize()
    (oracle/_ref, built -O3 -DNDEBUG -ffp-contract=off) when it travelled to this box, else the C restatement.
    Returns (json dict, (framebuffer, z-buffer, print_render_stats() line)): the frame the CPU rendered is the
    checker of the parity gate (no timing is reported unless the GPU frame matches it)."""
    from oracle import orc
    n = min(sample, clip.shape[0])
    desc = (f"all {n} triangles of the same frame" if n == clip.shape[0] else f"first {n} triangles of the same scene") + ", rasterize() loop only"
    if os.path.exists(orc.REF_HARNESS_FAST):
        from tinyrenderder_amd import scenes
        fb, z, line, secs = orc.run_reference(W, H, 3, scenes.init_viewport(0, 0, W, H), [(orc.FLAT, None, clip[:n], None, col[:n])],
                                              harness=orc.REF_HARNESS_FAST, with_time=True)
        kind = "reference"
    else:
        o = orc.Oracle(W, H, 3)
        t0 = time.perf_counter()
        o.draw(orc.FLAT, clip[:n], colors=col[:n])
        secs = time.perf_counter() - t0
        fb, z, line = o.fb, o.z, orc.format_stats_line(o.stats)
        kind = "port"
    return ({"value": n / secs, "unit": "triangles/s", "cores": 1, "kind": kind, "sample": desc,
             "seconds": round(secs, 3), "host_cpus": os.cpu_count()}, (fb, z, line))


def parity_gate(ctx, kind, dclip, dvary, dcol, uniforms, n, cpu_frame, checker, golden_name):
    """Render the first n triangles once more, outside the timed region, through exactly the calls of step(), and
    compare framebuffer bytes, z-buffer bits and the print_render_stats() line with the frame the CPU rendered
    (`cpu_frame` from cpu_baseline, or None) and with the committed digests of the reference's frame
    (tests/golden/golden_fullsize.json, when the workload is one of its cases)."""
    import hashlib
    ctx.reset_stats()
    ctx.clear()
    ctx.draw(kind, dclip[:n], varyings=None if dvary is None else dvary[:n], colors=None if dcol is None else dcol[:n],
             uniforms=uniforms, device=True)
    ctx.flush()
    fb, z, line = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats_line()
    out = {"checked": False, "triangles": int(n), "fb": None, "z": None, "stats": None, "against": [], "stats_line": line}
    ok = True
    if cpu_frame is not None:
        rfb, rz, rline = cpu_frame
        out["fb"] = bool(np.array_equal(fb, rfb))
        out["z"] = bool(np.array_equal(z.view(np.uint64), rz.view(np.uint64)))
        out["stats"] = bool(line == rline)
        out["against"].append(checker)
        out["checked"] = True
        ok = out["fb"] and out["z"] and out["stats"]
        if not ok:
            out["cpu_stats_line"] = rline
            out["fb_bytes_differing"] = int((fb != rfb).sum())
            out["z_values_differing"] = int((z.view(np.uint64) != rz.view(np.uint64)).sum())
    gpath = os.path.join(ROOT, "tests", "golden", "golden_fullsize.json")
    if golden_name and os.path.exists(gpath):
        g = json.load(open(gpath)).get(golden_name)
        if g:
            sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()
            gold = {"fb": sha(fb) == g["fb"], "z": sha(z) == g["z"], "stats": line == g["stats"]}
            out["golden"] = gold
            out["against"].append(f"tests/golden/golden_fullsize.json[{golden_name}] (digests of the compiled reference's frame)")
            out["checked"] = True
            ok = ok and all(gold.values())
    out["ok"] = bool(ok) if out["checked"] else None
    return out, fb


def end_to_end(ctx, kind, clip, vary, col, uniforms, frames=2):
    """PCIe-inclusive frame: the triangle stream handed over as pageable HOST arrays (trgl_draw copies them before it
    returns), the flush, and the read-back of the finished framebuffer.  Reported next to `value`, never as `value`."""
    ts = []
    for _ in range(frames + 1):
        t0 = time.perf_counter()
        ctx.clear()
        ctx.draw(kind, clip, varyings=vary, colors=col, uniforms=uniforms)
        ctx.flush()
        ctx.read_framebuffer()
        ts.append(time.perf_counter() - t0)
    dt = min(ts[1:])
    h2d = clip.nbytes + (0 if vary is None else vary.nbytes) + (0 if col is None else col.nbytes)
    return {"ms_per_frame": dt * 1e3, "triangles_per_s": clip.shape[0] / dt, "h2d_bytes": int(h2d),
            "d2h_bytes": ctx.width * ctx.height * ctx.bpp,
            "includes": "H2D of the clip-space stream (+ varyings/colours) from pageable host memory, flush, D2H of the framebuffer"}


def frames_in_flight(Context, W, H, kind, dclip, dcol, fb_expected, frames=20, depth=2):
    """Throughput with `depth` frames in flight: `depth` contexts, each on its own HIP stream, render the same frame in
    turn, so that the HBM-bound phases of one frame (k_setup, binning) may run beside the issue-bound k_raster of the
    previous one.  Same work per frame as a timed step (clear, draw, flush); every context's last frame is compared with
    the timed frame.  Reported next to `value`, never as `value`: a step of the contract is one frame start to finish."""
    import torch
    ctxs = [Context(W, H, 3) for _ in range(depth)]

    def begin(c):
        c.clear(); c.draw(kind, dclip, colors=dcol, device=True); c.flush_begin()
    for c in ctxs:
        begin(c); c.flush_end(); c.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = []
    for f in range(frames):
        if len(pending) == depth:
            pending.pop(0).flush_end()
        begin(ctxs[f % depth]); pending.append(ctxs[f % depth])
    for c in pending:
        c.flush_end()
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t0
    same = all(np.array_equal(c.read_framebuffer(), fb_expected) for c in ctxs) if fb_expected is not None else None
    for c in ctxs:
        c.close()
    return {"depth": depth, "frames": frames, "ms_per_frame": dt * 1e3 / frames, "triangles_per_s": dclip.shape[0] * frames / dt,
            "frames_equal_timed_frame": same}


def secondary_workload(name, dev, steps, warmup, check):
    """BASELINE configs[1] / [2] (PHONG head stand-in at 2048^2 with the diffuse map / at 4096^2 with all three maps) as a
    secondary record of the default line: frame time with profiling off, phases and the k_raster roofline entry from a profiled
    loop, parity of the rendered frame against the C restatement (the PHONG body is restated from main.cpp's text)."""
    import torch
    from tinyrenderder_amd import scenes
    from tinyrenderder_amd.api import Context, PHONG, make_uniforms, PHASE_RASTER, PHASE_SETUP, PHASE_BIN, PHASE_TOTAL, PHASE_RASTER_KERNEL
    W = H = 2048 if name == "c2" else 4096
    hd = scenes.head_standin(7, W, H)
    d_, n_, s_ = scenes.procedural_textures(1024)
    textures = {0: d_} if name == "c2" else {0: d_, 1: n_, 2: s_}
    slots = (0, -1, -1) if name == "c2" else (0, 1, 2)
    uniforms = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, *slots)
    N = hd["clip"].shape[0]
    dclip, dvary = torch.from_numpy(hd["clip"]).cuda(), torch.from_numpy(hd["varyings"]).cuda()
    with Context(W, H, 3, device=dev) as ctx:
        for slot, t in textures.items():
            ctx.upload_texture(slot, t)

        def step():
            ctx.clear(); ctx.draw(PHONG, dclip, varyings=dvary, uniforms=uniforms, device=True); ctx.flush()
        for _ in range(warmup):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        ctx.set_profiling(True); ctx.reset_phase_ms()
        for _ in range(max(3, steps // 4)):
            step()
        ph, nfl = ctx.phase_ms()
        ctx.set_profiling(False)
        raster_ms = ph[PHASE_RASTER_KERNEL] / max(nfl, 1)
        algo = W * H * 11 + N * (96 + 8 * 24)
        rec = {"workload": f"configs[{1 if name == 'c2' else 2}]: {N}-triangle head stand-in, {W}x{H}, PHONG, maps {sorted(textures)}",
               "ms_per_step": ms, "triangles_per_s": N / (ms * 1e-3),
               "phase_ms": {"setup": ph[PHASE_SETUP] / max(nfl, 1), "bin": ph[PHASE_BIN] / max(nfl, 1), "raster": ph[PHASE_RASTER] / max(nfl, 1),
                            "raster_kernel": raster_ms, "flush_total": ph[PHASE_TOTAL] / max(nfl, 1)},
               "roofline": {"bound": "hbm", "kernel": "k_raster<phong>", "achieved": algo / (raster_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": algo / (raster_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo, "avg_launch_ms": raster_ms,
                            "traffic": None}}
        if check:
            from oracle import orc
            o = orc.Oracle(W, H, 3)
            for slot, t in textures.items():
                o.upload_texture(slot, t)
            o.draw(orc.PHONG, hd["clip"], hd["varyings"], uniforms=orc.Uniforms.from_buffer_copy(bytes(uniforms)))
            fb, z, line = ctx.read_framebuffer(), ctx.read_zbuffer(), ctx.stats_line()
            ctx.reset_stats(); step()
            line = ctx.stats_line()
            rec["parity"] = {"checked": True, "fb": bool(np.array_equal(fb, o.fb)), "z": bool(np.array_equal(z.view(np.uint64), o.z.view(np.uint64))),
                             "stats": bool(line == orc.format_stats_line(o.stats)),
                             "against": ["oracle/trgl_oracle.c (C restatement; PHONG body restated from main.cpp text)"]}
            rec["parity"]["ok"] = bool(rec["parity"]["fb"] and rec["parity"]["z"] and rec["parity"]["stats"])
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--triangles", type=int, default=10_000_000)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="triangles timed on the host CPU (0 = skip)")
    ap.add_argument("--writeout-frames", type=int, default=20,
                    help="clear-only frames timed after the run for the write-out figure (0 = skip, e.g. under rocprofv3 so "
                         "that the k_raster statistics hold the full launches only)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gate (profiling runs)")
    ap.add_argument("--end-to-end-frames", type=int, default=2, help="PCIe-inclusive frames timed after the run (0 = skip)")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="after the run: throughput with this many frames in flight on separate streams (reported beside value; 1 = skip)")
    ap.add_argument("--partition", default="strips", choices=["strips", "bands"],
                    help="N > 1: one horizontal strip per rank (default), or bands of --band-rows rows dealt round-robin (load-balanced)")
    ap.add_argument("--band-rows", type=int, default=128)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, the measured configuration) or gloo: a rehearsal of the N > 1 code path with all ranks "
                         "on ONE GPU (a 1-GPU box cannot host two RCCL ranks); its numbers mean nothing")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL strip-gather path even with one rank (rehearsal)")
    ap.add_argument("--secondary", default="c2,c3",
                    help="secondary workloads appended to the default line as records (one GPU, default workload only; '' = none)")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps of the second, profiled loop (phase times, k_raster launch duration)")
    ap.add_argument("--workload", default="c4", choices=["c4", "c2", "c3"],
                    help="c4 (default, the metric's config): 10 M random flat triangles; c2/c3: PHONG head stand-in at 2048/4096")
    args = ap.parse_args()
    exit_code = [0]

    import torch
    from tinyrenderder_amd import scenes, shard
    from tinyrenderder_amd.api import Context, FLAT, PHASE_RASTER, PHASE_SETUP, PHASE_BIN, PHASE_TOTAL, PHASE_RASTER_KERNEL

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    dev = local_rank % max(torch.cuda.device_count(), 1) if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from tinyrenderder_amd.api import PHONG, make_uniforms
    W = H = args.size
    N = args.triangles
    kind, uniforms, dvary, K, textures = FLAT, None, None, 0, {}
    if args.workload == "c4":
        clip, col = scenes.random_triangles(N, W, H)                    # same seed on every rank
        dcol = torch.from_numpy(col.view(np.int32)).cuda()
        wl_name = f"configs[3]: {N} synthetic random triangles, {W}x{H} RGB + fp64 z, flat shader"
    else:   # secondary workloads (not the headline metric): BASELINE configs[1]/[2], PHONG on the head stand-in
        W = H = 2048 if args.workload == "c2" else 4096
        hd = scenes.head_standin(7, W, H)
        clip, col, dcol, N, kind, K = hd["clip"], None, None, hd["clip"].shape[0], PHONG, 24
        d_, n_, s_ = scenes.procedural_textures(1024)
        textures = {0: d_} if args.workload == "c2" else {0: d_, 1: n_, 2: s_}
        slots = (0, -1, -1) if args.workload == "c2" else (0, 1, 2)
        uniforms = make_uniforms(hd["model_view"], hd["key"], hd["fill"], hd["rim"], 1.0, *slots)
        dvary = torch.from_numpy(hd["varyings"]).cuda()
        wl_name = f"configs[{1 if args.workload == 'c2' else 2}]: {N}-triangle head stand-in, {W}x{H}, PHONG, maps {sorted(textures)}"
    dclip = torch.from_numpy(clip).cuda()

    ctx = Context(W, H, 3, device=dev)
    for slot, t in textures.items():
        ctx.upload_texture(slot, t)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)          # order with torch / RCCL on one stream
    y0, y1 = shard.strip_rows(H, world, rank)
    full_fb = None
    if use_dist:
        if args.partition == "bands":
            ctx.set_interleave(args.band_rows, rank, world)
        else:
            ctx.set_strip(y0, y1)
        full_fb = shard.framebuffer_tensor(ctx)

    def submit(c):
        c.clear()
        c.draw(kind, dclip, varyings=dvary, colors=dcol, uniforms=uniforms, device=True)

    # N > 1: join the colour strips so that every rank ends with the whole TGAImage buffer.  The gather runs on RCCL's stream;
    # only the raster half of the NEXT frame touches the framebuffer, so that frame's setup and binning run under it
    # (shard.StripLoop; tests/test_gpu_parity.py drives the same loop with two contexts on one GPU).
    if args.partition == "bands":
        start_gather = lambda: shard.gather_bands(full_fb, W, H, 3, args.band_rows, rank, world, async_op=True)
    else:
        start_gather = lambda: shard.gather_strips(full_fb, W, H, 3, rank, world, async_op=True)
    loop = shard.StripLoop(ctx, start_gather) if use_dist else None

    def step():
        if loop is None:
            submit(ctx)
            ctx.flush()
        else:
            loop.step(submit)

    def fence():
        if use_dist:
            loop.finish()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # the headline loop: event profiling OFF, so that no flush waits for the events of the one before it
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # a second, short loop with profiling ON: HIP events on the context's stream around the phases and right around the k_raster launch
    ctx.set_profiling(True)
    ctx.reset_phase_ms()
    tp0 = time.perf_counter()
    for _ in range(max(args.profile_steps, 1)):
        step()
    fence()
    elapsed_profiled = time.perf_counter() - tp0
    phase_ms, nfl = ctx.phase_ms()
    rank_phases = None
    if use_dist:      # every rank's phases, for the record of the multi-GPU run
        mine = torch.tensor([phase_ms[PHASE_SETUP], phase_ms[PHASE_BIN], phase_ms[PHASE_RASTER], phase_ms[PHASE_RASTER_KERNEL], phase_ms[PHASE_TOTAL]],
                            dtype=torch.float64, device="cuda") / max(nfl, 1)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        rank_phases = [dict(zip(("setup", "bin", "raster", "raster_kernel", "flush_total"), [float(v) for v in p.cpu()])) for p in allp]
    info = ctx.last_flush_info()
    fb_timed = ctx.read_framebuffer() if (rank == 0 and world == 1 and not args.no_parity) else None     # the last timed frame
    # frame write-out alone (outside the timed region): clear + flush with no triangles = k_raster storing W*H*(8+bpp)
    # bytes once, straight from the clear values.  It is the ceiling of the tile-out path, NOT what the C4 launch achieves
    # on the same bytes (see "write_path" below).
    writeout = None
    if world == 1 and args.writeout_frames > 0:
        ctx.reset_phase_ms()
        for _ in range(args.writeout_frames):
            ctx.clear()
            ctx.flush()
        wo_ms, wo_n = ctx.phase_ms()
        wo_us = wo_ms[PHASE_RASTER_KERNEL] / max(wo_n, 1) * 1e3
        writeout = {"kind": "clear-only", "bytes": W * H * 11, "kernel": "k_raster on a cleared frame without triangles", "avg_launch_us": wo_us,
                    "achieved": W * H * 11 / (wo_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": W * H * 11 / (wo_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    ctx.set_profiling(False)

    # ---- parity gate: no timing is reported unless the frame that was timed is the reference's ----
    parity, cpu, e2e, inflight = {"checked": False, "ok": None}, None, None, None
    golden_name = "c4_4096_10m" if (args.workload == "c4" and W == 4096 and N == 10_000_000) else None
    if rank == 0 and world == 1 and not args.no_parity:
        cpu_frame, checker, n_par = None, None, N
        if args.workload == "c4" and args.cpu_sample > 0:
            cpu, cpu_frame = cpu_baseline(clip, col, W, H, args.cpu_sample)
            n_par = min(args.cpu_sample, N)
            checker = ("the reference's own rasterize() (oracle/_ref/ref_harness_fast)" if cpu["kind"] == "reference"
                       else "oracle/trgl_oracle.c (C restatement)")
        elif args.workload != "c4" and args.cpu_sample > 0:    # PHONG stand-in frames: the C restatement is the only CPU checker
            from oracle import orc
            o = orc.Oracle(W, H, 3)
            for slot, t in textures.items():
                o.upload_texture(slot, t)
            o.draw(orc.PHONG, clip, hd["varyings"], uniforms=orc.Uniforms.from_buffer_copy(bytes(uniforms)))
            cpu_frame, checker = (o.fb, o.z, orc.format_stats_line(o.stats)), "oracle/trgl_oracle.c (C restatement; PHONG body restated from main.cpp text)"
        parity, fb_par = parity_gate(ctx, kind, dclip, dvary, dcol, uniforms, n_par, cpu_frame, checker, golden_name)
        if n_par == N:
            parity["timed_frame_equals_parity_frame"] = bool(np.array_equal(fb_timed, fb_par))
            if parity["checked"]:
                parity["ok"] = bool(parity["ok"] and parity["timed_frame_equals_parity_frame"])
        if args.end_to_end_frames > 0:
            e2e = end_to_end(ctx, kind, clip, None if dvary is None else hd["varyings"], col, uniforms, args.end_to_end_frames)
        if args.frames_in_flight > 1 and kind == FLAT:
            inflight = frames_in_flight(Context, W, H, kind, dclip, dcol, fb_timed, frames=args.steps, depth=args.frames_in_flight)
    elif rank == 0 and world > 1 and golden_name and not args.no_parity:
        # multi-rank: rank 0 holds the gathered framebuffer (colour strips only travel); compare it with the reference's digest
        import hashlib
        gpath = os.path.join(ROOT, "tests", "golden", "golden_fullsize.json")
        if os.path.exists(gpath):
            g = json.load(open(gpath))[golden_name]
            sha = hashlib.sha256(full_fb.cpu().numpy().tobytes()).hexdigest()
            parity = {"checked": True, "ok": sha == g["fb"], "fb": sha == g["fb"], "z": None, "stats": None,
                      "against": [f"tests/golden/golden_fullsize.json[{golden_name}] (gathered framebuffer on rank 0)"]}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        tri_per_s = N * args.steps / elapsed
        # dominant kernel = the tile raster; algorithmic bytes per launch: every pixel's colour and z
        # leave the chip once (W*H*11 B) and every triangle's clip-space vertices are consumed once (N*96 B).
        algo_bytes = (W * H * 11 + N * (96 + 8 * K)) / world
        raster_ms = phase_ms[PHASE_RASTER_KERNEL] / max(nfl, 1)       # the k_raster launch alone, HIP events on its stream
        achieved = algo_bytes / (raster_ms * 1e-3) / 1e9
        traffic, write_path = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("workload") == f"c4_{W}x{H}_{N}":
                traffic = tj.get("raster_hbm_bytes_per_launch")
                # what the REAL launch does with the W*H*11 B that have to leave the chip (PMC WRITE_SIZE of k_raster)
                write_path = {"kind": "full C4 launch", "write_size_bytes": tj["write_size_kib"] * 1024.0, "algorithmic_bytes": W * H * 11,
                              "amplification": tj["write_size_kib"] * 1024.0 / (W * H * 11),
                              "source": "profiles/traffic.json (rocprofv3 --pmc WRITE_SIZE of the same command, " + tj.get("round", "r01") + ")"}
        failed = parity["checked"] and not parity["ok"]
        out = {
            "metric": "triangles/sec (+ Mpixels/sec) at 4096x4096; achieved HBM GB/s vs peak",
            "value": None if failed else tri_per_s, "unit": "triangles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_profiled": elapsed_profiled * 1e3 / max(args.profile_steps, 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_name,
                       "width": W, "height": H, "triangles": N, "tile": 32,
                       "parallelism": (f"screen strips x{world}" if args.partition == "strips" else f"interleaved {args.band_rows}-row bands x{world}")
                                      + (" + RCCL all-gather of colour strips" if world > 1 else "")},
            "rccl_ranks": (dist.get_world_size() if (use_dist and args.backend == "nccl") else 0),
            "parity": parity,
            "mpixels_per_s": W * H * args.steps / elapsed / 1e6,
            "tri_tile_pairs": info["pairs"],
            "phase_ms": {"setup": phase_ms[PHASE_SETUP] / max(nfl, 1), "bin": phase_ms[PHASE_BIN] / max(nfl, 1),
                         "raster": phase_ms[PHASE_RASTER] / max(nfl, 1), "raster_kernel": raster_ms,
                         "flush_total": phase_ms[PHASE_TOTAL] / max(nfl, 1)},
            "roofline": {"bound": "hbm", "kernel": "k_raster<flat>" if kind == FLAT else "k_raster<phong>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": raster_ms},
        }
        # the two streaming phases next to the raster kernel, as bytes moved per second (HIP-event phase times), on the bytes the
        # IMPLEMENTATION moves and on SURVEY.md section 8(d)'s ALGORITHMIC bytes:
        # k_setup reads 96 B of vertices + 4 B of colour and writes a 128-B record + 12 B per triangle (algorithmic: the 100 B read);
        # binning moves 46 B per pair with 16-bit tile keys - a pair is (tile 2 B, triangle 4 B, 4x4 block mask 2 B): expand writes 8, two
        # histogram reads 4, two scatter passes 32, bounds 2 (62 B with the 32-bit keys of frames beyond 65536 tiles) - + 12 B per
        # triangle (algorithmic: one 4-B list entry per pair written once and read once by the raster)
        setup_ms = phase_ms[PHASE_SETUP] / max(nfl, 1)
        bin_ms = phase_ms[PHASE_BIN] / max(nfl, 1)
        if kind == FLAT and setup_ms > 0 and bin_ms > 0:
            setup_bytes = N * (96 + 4 + 12) + (N / world) * 128
            bin_bytes = info["pairs"] * (46 if ((W + 31) // 32) * ((H + 31) // 32) <= 65536 else 62) + N * 12
            gbs = lambda b, ms: b / (ms * 1e-3) / 1e9
            out["streaming_phases"] = {
                "k_setup": {"bytes": setup_bytes, "ms": setup_ms, "achieved": gbs(setup_bytes, setup_ms), "unit": "GB/s",
                            "frac": gbs(setup_bytes, setup_ms) / HBM_PEAK_GBS, "algorithmic_bytes": N * 100,
                            "algorithmic_frac": gbs(N * 100, setup_ms) / HBM_PEAK_GBS},
                "binning": {"bytes": bin_bytes, "ms": bin_ms, "achieved": gbs(bin_bytes, bin_ms), "unit": "GB/s",
                            "frac": gbs(bin_bytes, bin_ms) / HBM_PEAK_GBS, "algorithmic_bytes": info["pairs"] * 8,
                            "algorithmic_frac": gbs(info["pairs"] * 8, bin_ms) / HBM_PEAK_GBS}}
        if rank_phases:
            out["rank_phase_ms"] = rank_phases
        if writeout:
            out["writeout"] = writeout
        if write_path:
            out["write_path"] = write_path
        if e2e:
            out["end_to_end"] = e2e
        if inflight:
            out["frames_in_flight"] = inflight
        if cpu:
            out["cpu_baseline"] = cpu
        if world == 1 and args.workload == "c4" and args.secondary:
            out["secondary"] = [secondary_workload(nm, dev, 40, 5, check=not args.no_parity and args.cpu_sample > 0) for nm in args.secondary.split(",") if nm]
        print(json.dumps(out))
        if failed:
            print("PARITY FAILURE: the GPU frame differs from the reference's; no throughput is reported", file=sys.stderr)
            exit_code[0] = 1
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if exit_code[0]:
        sys.exit(exit_code[0])


if __name__ == "__main__":
    main()
