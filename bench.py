"""Headline benchmark: point-clouds/sec, forward+backward(+optimizer step), ModelNet40-shaped input
(1024 points, k=20, batch 32 per GPU), synthetic data, random-init weights, fp32.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default = BASELINE configuration 2 (the one the metric is quoted on), weak scaling: 32 clouds per GPU.  The other
configurations (deltaconv_amd/configs.py) and the STRONG-scaling form of a configuration -- BASELINE configs 4 / 5 are
defined as a global batch of 16 / 8 clouds data-parallel over 8 GPUs -- are selected by flags:

    python bench.py --config C4 --global-batch 16 --gpus 8      # 2 clouds per rank, "scaling": "strong", BatchNorm
                                                                 # statistics over the global batch (= one process on 16)
    python bench.py --config C5                                  # the whole configuration on one GPU

One JSON line on rank 0 (contract in the task statement) carrying `roofline` (the sparse operator
apply, measured live with HIP events) and `cpu_baseline` (the oracle = CPU port of the reference
path, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"   # before HIP starts (deltaconv_amd/graph_step.py)

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE.json configuration (deltaconv_amd/configs.py); C2 = the one the metric is quoted on")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="STRONG scaling: this many clouds in total, global-batch / world per rank, BatchNorm statistics over the "
                         "global batch (BASELINE configs 4 / 5: --config C4 --global-batch 16, --config C5 --global-batch 8)")
    ap.add_argument("--batch", type=int, default=None, help="clouds per GPU (weak scaling; default: the configuration's batch)")
    ap.add_argument("--points", type=int, default=None, help="points per cloud (default: the configuration's)")
    ap.add_argument("--k", type=int, default=None, help="neighbours (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-chain", action="store_true",
                    help="skip the second timing of the same step with every dense product on the exact fp32 MFMA chain")
    ap.add_argument("--cpu-clouds", type=int, default=None, help="clouds in the CPU-baseline sample (default 8 / 4 / 4 / 2 for C2 .. C5)")
    ap.add_argument("--no-cpu-full-batch", action="store_true",
                    help="skip the second CPU-baseline entry on the full batch of the configuration (~25 s)")
    ap.add_argument("--no-in-step-stamps", action="store_true",
                    help="skip the second capture of the step with device-clock stamps (kernel-trace runs: the stamp replays would be "
                         "the last steps of the trace); `roofline.frac` is then the rotating-buffer number")
    ap.add_argument("--no-graph", action="store_true",
                    help="enqueue every kernel from Python each step instead of replaying the captured HIP graph")
    ap.add_argument("--resident-batches", type=int, default=4, help="distinct synthetic batches cycled through")
    ap.add_argument("--sync-bn", action="store_true", default=None,
                    help="BatchNorm statistics over the global batch (all-reduced fp64 sums); the data-parallel step is then ONE "
                         "graph with its collectives captured.  Default: on with --global-batch (strong scaling), off otherwise")
    ap.add_argument("--no-sync-bn", dest="sync_bn", action="store_false")
    ap.add_argument("--spawn", action="store_true",
                    help="go through the self-launcher even with --gpus 1 (exercises the path `--gpus N > 1` takes; tests/test_gpu_dist.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gradient all-reduce path even with one rank (self-test)")
    args = ap.parse_args(argv)
    from deltaconv_amd.configs import CONFIGS
    cfg = CONFIGS[args.config]
    args.points = cfg["N"] if args.points is None else args.points
    args.k = cfg["k"] if args.k is None else args.k
    args.strong = args.global_batch is not None
    if args.sync_bn is None:
        args.sync_bn = args.strong
    if args.cpu_clouds is None:
        args.cpu_clouds = {"C2": 8, "C3": 4, "C4": 4, "C5": 2}[args.config]
    if args.config != "C2":
        args.no_cpu_full_batch = True       # the bounded sample only: a full C3 .. C5 batch is minutes of CPU work
    return args


def per_rank_clouds(args, world):
    """Clouds of one rank: global-batch / world (strong scaling: fixed total work) or the configuration's batch per GPU (weak)."""
    from deltaconv_amd.configs import CONFIGS
    if args.strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} ranks")
        return args.global_batch // world
    return CONFIGS[args.config]["B"] if args.batch is None else args.batch


def _capture_mode():
    """With a process group alive its watchdog thread may query events while this thread captures: a global-mode capture would
    be invalidated by that call (deltaconv_amd/graph_step.py)."""
    return "thread_local" if dist.is_initialized() else "global"


def apply_roofline(graph, grad, div, C, iters=200):
    """Live measurement of the dominant kernel family -- the ELL operator applies -- with HIP events on
    the stream the kernels are launched on (torch's current stream; the C ABI launches there).
    Algorithmic bytes per launch: input once + output once +
    ids/coefficients once:  grad/div 12*C*Nt + 12*E,  div|curl|norm 20*C*Nt + 12*E,  hodge 16*C*Nt + 12*E.
    The headline kernel is the fused div|curl|norm apply (largest forward apply of every DeltaConv
    layer); the other members of the family are listed beside it."""
    from deltaconv_amd._lib import lib
    n, k = graph.n, graph.k
    dev = grad.coef.device
    E = n * k
    x = torch.randn(n, C, device=dev)
    v = torch.randn(2 * n, C, device=dev)
    dcn = torch.randn(n, 3 * C, device=dev)
    y1 = torch.empty(n, C, device=dev)
    y2 = torch.empty(2 * n, C, device=dev)
    y3 = torch.empty(n, 3 * C, device=dev)
    from deltaconv_amd import _ops
    from deltaconv_amd.geometry import graph as _G
    arg = torch.empty(n, C, dtype=torch.uint8, device=dev)
    # the product's dispatch (deltaconv_amd/_ops.py): from the graph's tile plan (neighbour rows in LDS, csrc/ell_tile.h)
    # when it applies, else through the gather path
    cases_F = {
        "div_curl_norm": (lambda: _ops.fwd_apply("div_curl_norm", div, v, C, C, y3, 3 * C), 20 * C * n + 12 * E),
        "grad": (lambda: _ops.fwd_apply("grad", grad, x, C, C, y2, C), 12 * C * n + 12 * E),
        "div": (lambda: _ops.fwd_apply("div", div, v, C, C, y1, C), 12 * C * n + 12 * E),
        "hodge": (lambda: _ops.fwd_apply("hodge", grad, dcn, C, 3 * C, y2, C), 16 * C * n + 12 * E),
        "knn_max": (lambda: _ops.fwd_knn_max(graph, x, C, C, y1, C, arg), 9 * C * n + 4 * E),
    }

    # backward half of the operator (round 4): the transposed applies and the max-aggregation backward with the operand
    # layouts of the layer node (accumulating outputs are read once more): from the transposed tile plan
    # (csrc/ell_tileT.h) when the graph has one, else over the CSC through the gather path
    dyv, a1 = torch.randn(2 * n, C, device=dev), torch.randn(n, C, device=dev)
    o1, o2c, dvv = torch.empty(n, C, device=dev), torch.zeros(n, 2 * C, device=dev), torch.zeros(2 * n, C, device=dev)
    argr = torch.randint(0, k, (n, C), device=dev).to(torch.uint8)
    cases_T = {
        "div_curl_norm_T": (lambda: _ops.bwd_div_curl_norm(div, dcn, C, 3 * C, v, C, dvv, C, 1), 36 * C * n + 12 * E),
        "hodge_T": (lambda: _ops.bwd_apply("hodge", grad, dyv, C, C, o2c, 2 * C, 1), 24 * C * n + 12 * E),
        "grad_T_sum": (lambda: _ops.bwd_grad_sum(grad, dyv, C, C, a1, C, None, 0, o1, C), 16 * C * n + 12 * E),
        "knn_max_bwd": (lambda: _ops.bwd_knn_max(graph, argr, x, C, C, o1, C, 0), 9 * C * n + 4 * E),
    }

    def measure(passes=2, cases=None):
        cases = cases or cases_F
        # two passes over the family, the second one reported: the first replay series of a case in a process runs
        # 1 - 1.5 us slower than every later one (r03 lab, tools/archive_r01_r04.tar.gz:archive/tile_upw.py: 14.6 then 13.1 x 7 for the fused apply --
        # fresh allocations / cold translation caches), and the training step launches these kernels every iteration
        out = {}
        for rep in range(passes):
            first = {k_: v_["us"] for k_, v_ in out.items()}
            _measure_pass(out, cases)
        for k_ in out:
            out[k_]["first_pass_us"] = first.get(k_)
        return out

    def _measure_pass(out, cases):
        for name, (fn, nbytes) in cases.items():
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            # back-to-back launches replayed from a captured HIP graph: the Python / ctypes enqueue of one call (~10 us)
            # is as long as these kernels, eager launches would time the host
            per = 50
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=_capture_mode()):
                for _ in range(per):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(max(1, iters // per)):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / (max(1, iters // per) * per) * 1e-3
            out[name] = dict(us=round(t * 1e6, 2), bytes=nbytes, GBs=round(nbytes / t / 1e9, 1),
                             frac=round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4))

    tiled = graph.tile_plan() is not None
    fam = measure()

    # Dispatch overhead of one launch of the graded kernel: HIP events around back-to-back launches (what rocprofv3's kernel
    # duration corresponds to: dispatch ramp + execution + end-of-kernel release) minus the device-clock stamps of the SAME
    # launches (first workgroup entry -> last workgroup exit).  Added to the in-step stamp durations further down so that the
    # graded `us_per_launch` is the quantity a kernel trace of the step shows.
    def _dispatch_overhead(fn, event_us, per=50, reps=8):
        if not tiled:
            return None
        stamps = torch.zeros(per, 4, dtype=torch.int64, device=dev)
        rawb = lib.raw("dc_stamp_buffer")
        rawb(stamps.data_ptr(), per)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=_capture_mode()):
                for _ in range(per):
                    fn()
            used = lib.raw("dc_stamp_count")()
        finally:
            rawb(None, 0)
        if used != per:
            return None
        meds, evs = [], []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(reps):
            stamps[:, 0] = 2 ** 62
            stamps[:, 1] = 0
            torch.cuda.synchronize()
            e0.record()
            g.replay()                       # the SAME launches by both clocks: events around the replay, stamps inside it
            e1.record()
            torch.cuda.synchronize()
            rec = stamps.cpu()
            meds.append(float(((rec[:, 1] - rec[:, 0]).double() * 1e-2).median()))
            evs.append(e0.elapsed_time(e1) * 1e3 / per)
        exec_us, ev_us = sorted(meds)[len(meds) // 2], sorted(evs)[len(evs) // 2]
        return dict(execution_us=round(exec_us, 2), events_us=round(ev_us, 2), overhead_us=round(max(0.0, ev_us - exec_us), 2))
    dispatch = _dispatch_overhead(cases_F["div_curl_norm"][0], fam["div_curl_norm"]["us"])
    fam_gather = None
    if tiled:                                   # the same applies through the gather path (A/B, same results)
        plan, graph._tile_plan = graph._tile_plan, False
        fam_gather = measure()
        graph._tile_plan = plan
    # The headline kernel on ROTATING buffers: 12 (input, output) sets = 600 MB cycled through, so no launch finds its
    # operands in the 256 MB Infinity Cache or the L2s -- the condition inside the training step, where every apply reads
    # what another kernel wrote long before (the back-to-back replay above re-reads the same 50 MB: cache_level).
    def _rotating(make_call, nbytes, sets=12, rounds=4):
        calls = [make_call() for _ in range(sets)]
        for c in calls:
            c()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=_capture_mode()):
            for _ in range(rounds):
                for c in calls:
                    c()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / (4 * rounds * sets) * 1e-3
        return dict(us=round(t * 1e6, 2), bytes=nbytes, GBs=round(nbytes / t / 1e9, 1), frac=round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
                    sets=sets, footprint_MB=round(sets * nbytes / 1e6))

    def _mk_dcn():
        vv, oo = torch.randn(2 * n, C, device=dev), torch.empty(n, 3 * C, device=dev)
        return lambda: _ops.fwd_apply("div_curl_norm", div, vv, C, C, oo, 3 * C)
    in_step = _rotating(_mk_dcn, 20 * C * n + 12 * E)
    tiled_T = graph.tile_plan_T() is not None
    if tiled_T:
        grad.coefTt(), div.coefTt()
    graph.csc(), grad.coefT(), div.coefT()
    fam_T = measure(cases=cases_T)
    fam_T_gather = None
    if tiled_T:
        plan_T, graph._tile_plan_T = graph._tile_plan_T, False
        fam_T_gather = measure(cases=cases_T)
        graph._tile_plan_T = plan_T
    # the hand-written fp32-MFMA GEMMs (csrc/gemm.hip, gemm_tn.hip): forward product of the embedding MLP with the
    # BatchNorm-statistics epilogue (the largest GEMM of the step) and the layer-2 v_mlp weight gradient
    def _time(fn, it=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e-3
    Me, Ne, Ke = n, 1024, 512
    Xe, We = torch.randn(Me, Ke, device=dev), torch.nn.Parameter(torch.randn(Ne, Ke, device=dev))
    from deltaconv_amd.nn import fused as _fused
    Ye, coef = torch.empty(Me, Ne, device=dev), torch.empty(4, Ne, device=dev)
    ge = torch.ones(Ne, device=dev)
    nbe = lib.raw("dc_linear_stats_workspace_bytes")(Me, Ne, Ke, 0)
    wse = torch.empty((nbe + 7) // 8, dtype=torch.float64, device=dev)
    def _embed():       # as the model runs it: the weight operand from its pre-split bf16 planes (nn/fused.py)
        with torch.no_grad():
            _fused._hint_planes(We, False)
            lib.call("dc_linear_bn_stats_forward", Xe, Ke, We, Ke, Me, Ne, Ke, Ye, Ne, ge, ge, 1e-5, 0.1, None, None, coef[0], coef[1],
                     coef[2], coef[3], 0, wse, nbe)
    te = _time(_embed)
    R, M, N = 2 * n, 256, 256
    A, Bm = torch.randn(R, M, device=dev), torch.randn(R, N, device=dev)
    Cout = torch.empty(M, N, device=dev)
    nb = lib.raw("dc_gemm_tn_workspace_bytes")(R, M, N)
    ws = torch.empty((nb + 3) // 4, device=dev)
    tg = _time(lambda: lib.call("dc_gemm_tn", A, M, Bm, N, R, M, N, Cout, N, 0, ws, ws.numel() * 4))
    # Dense products: by default an fp32 product is six bf16 partial products (3 planes per operand, csrc/gemm.hip) on the
    # bf16 matrix pipe, fp32 accumulation -- error against fp64 at or below the exact fp32 MFMA chain's
    # (tests/test_gpu_gemm.py::test_split_products_no_worse_than_exact_chain).  The matrix pipe therefore executes 6 x the
    # algorithmic flops; `frac` prices THAT against the dense bf16 peak, `fp32_equivalent` is the algorithmic rate (the
    # exact chain's ceiling is the 157.3 TFLOP/s fp32 MFMA peak; DC_GEMM_EXACT=1 selects it).
    exact = os.environ.get("DC_GEMM_EXACT", "0") not in ("", "0")
    mult, peak = (1.0, 157.3) if exact else (6.0, 2500.0)
    path = ("exact fp32 chain v_mfma_f32_32x32x2_f32" if exact else
            "bf16 split products 6 x v_mfma_f32_32x32x16_bf16, fp32 accumulate, weight operand from pre-split planes")

    def _rate(flops, t):
        return dict(us=round(t * 1e6, 1), fp32_equivalent=round(flops / t / 1e12, 1), achieved=round(mult * flops / t / 1e12, 1),
                    frac=round(mult * flops / t / 1e12 / peak, 3))
    mfma = dict(kernel=f"gemm_kernel<128,128> + statistics epilogue + finaliser (Y = X W^T, {Me}x{Ne}x{Ke}, {path}, LDS-staged)",
                peak=peak, unit="TFLOP/s", pipe_flops_per_algorithmic_flop=mult, **_rate(2.0 * Me * Ne * Ke, te),
                weight_gradient=dict(kernel=f"gemm_kernel<128,128, A_KM, B_KN> over row slabs + ordered reduce (dW = dY^T X, {R}x{M}x{N})",
                                     **_rate(2.0 * R * M * N, tg)))
    head = fam["div_curl_norm"]
    # HBM bytes per launch from the PMC passes: counters cannot be read from inside the process, they come from separate
    # `rocprofv3 --pmc` runs of tools/pmc_apply.sh on the binary named by `traffic_tag` (profiles/README.md)
    traffic = traffic_tag = None
    try:
        import json as _json
        pmc = _json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_divcurlnorm.json")))
        if C == 64 and n == 32768 and k == 20 and tiled:      # the counters were taken on the tiled kernel
            traffic, traffic_tag = pmc["traffic_bytes"], pmc.get("tag")
    except Exception:
        pass
    # Second bound, the one that actually limits the kernel: every neighbour row is gathered
    # through the vector-memory (texture addresser / L1) path, 64 B/clk/CU.  Gathered bytes = 2 rows x 4C bytes per edge.
    gathered = 2 * 4 * C * E
    gh = (fam_gather or fam)["div_curl_norm"]           # the gather-path kernel this bound applies to
    l1_peak = 64.0 * 256 * 2.4e9 / 1e9                      # GB/s: 64 B/clk/CU x 256 CUs x 2.4 GHz
    l1 = dict(gathered_bytes=gathered, achieved=round(gathered / (gh["us"] * 1e-6) / 1e9, 1), peak=round(l1_peak, 1),
              unit="GB/s", frac=round(gathered / (gh["us"] * 1e-6) / 1e9 / l1_peak, 4),
              note="gather-path kernel (dc_apply_div_curl_norm); the tiled kernel moves ~1/7 of these bytes through this path")
    # the graded numbers are the ones of the kernel as it runs INSIDE the step (operands from HBM: rotating buffer sets); the
    # back-to-back replay on one 50 MB working set (Infinity-Cache resident) is reported beside them (round-4 verdict, item 4)
    return dict(bound="hbm", achieved=in_step["GBs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=in_step["frac"],
                frac_l3_resident=head["frac"], achieved_l3_resident=head["GBs"], us_per_launch_l3_resident=head["us"],
                traffic=traffic, traffic_tag=traffic_tag,
                cache_level="`achieved` / `frac`: 12 rotating (input, output) sets = 600 MB, operands come from HBM as inside the "
                            "training step; `*_l3_resident`: back-to-back replay on ONE set (~50 MB < the 256 MB Infinity Cache)",
                l1_gather=l1,
                kernel=("tile_fwd_kernel<DivCurlNormB> (dc_apply_div_curl_norm_tiled: fused ELL SpMM, neighbour rows in LDS "
                        "from the per-batch tile plan)" if tiled else
                        "divcurlnorm_fwd (dc_apply_div_curl_norm, fused ELL SpMM)"), channels=C,
                bytes_per_launch=head["bytes"], us_per_launch=in_step["us"], family=fam, family_gather_path=fam_gather,
                in_step=in_step, dispatch=dispatch,
                in_step_note=("the headline kernel on 12 rotating (input, output) sets (600 MB) = `achieved` / `frac` / `us_per_launch` "
                              "above; the `family*` blocks are back-to-back replays on one set each (Infinity-Cache resident)"),
                family_T=fam_T, family_T_gather_path=fam_T_gather,
                family_T_note=("backward half: transposed applies + max-aggregation backward at the layer node's operand layouts "
                               "(accumulating outputs counted read + written), " +
                               ("tileT_kernel from the transposed tile plan (source rows in LDS)" if tiled_T else "gather path over the CSC")),
                mfma=mfma)



STAMP_KINDS = {1: "div_curl_norm", 2: "hodge", 3: "div", 11: "div_curl_norm_T", 12: "hodge_T", 13: "grad_T_sum", 14: "knn_max_bwd",
               15: "div_T", 16: "edge_mlp_bwd"}
STAMP_BYTES = {1: lambda c, n, e: 20 * c * n + 12 * e, 2: lambda c, n, e: 16 * c * n + 12 * e, 3: lambda c, n, e: 12 * c * n + 12 * e,
               11: lambda c, n, e: 36 * c * n + 12 * e, 12: lambda c, n, e: 24 * c * n + 12 * e, 13: lambda c, n, e: 16 * c * n + 12 * e,
               14: lambda c, n, e: 9 * c * n + 4 * e}


def in_step_stamps(model, calc_loss, static, opt, n, k, replays=12):
    """Durations of the tiled operator kernels INSIDE the replayed training step, from device-clock stamps (csrc/common.h:
    dc_stamp_in / dc_stamp_out -- earliest workgroup entry to latest workgroup exit with its stores complete, 100 MHz constant
    clock): a SECOND capture of the same step with stamping armed (the timed step above never carries the stamps), replayed
    `replays` times with the records reset in between; per kernel instance the median over the replays.
    -> {name: [dict(C, us, bytes, frac) per instance in launch order]}"""
    from deltaconv_amd._lib import lib
    from deltaconv_amd.graph_step import GraphedTrainStep
    slots = 2048
    dev = static.pos.device
    stamps = torch.zeros(slots, 4, dtype=torch.int64, device=dev)
    raw = lib.raw("dc_stamp_buffer")
    raw(stamps.data_ptr(), slots)
    try:
        g = GraphedTrainStep(model, calc_loss, static, optimizer=opt, warmup=1)
    finally:
        used = lib.raw("dc_stamp_count")()
        tags = [lib.raw("dc_stamp_tag")(i) for i in range(used)]
        raw(None, 0)
    per = used // 2                       # one eager warm-up step + the captured one: the capture took the last `per` records
    if per == 0 or used != 2 * per:
        return None
    first = used - per
    samples = []
    for _ in range(replays):
        stamps[:, 0] = 2 ** 62
        stamps[:, 1] = 0
        g()
        torch.cuda.synchronize()
        rec = stamps[first:used].cpu()
        samples.append((rec[:, 1] - rec[:, 0]).double() * 1e-2)          # 100 MHz ticks -> us
    med = torch.stack(samples).median(0).values.tolist()
    out = {}
    e = n * k
    for tag, us in zip(tags[first:used], med):
        kind, c = tag // 1000, tag % 1000
        nbytes = STAMP_BYTES[kind](c, n, e) if kind in STAMP_BYTES else None
        out.setdefault(STAMP_KINDS.get(kind, str(kind)), []).append(
            dict(C=c, us=round(us, 2), bytes=nbytes, frac=None if nbytes is None else round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)))
    del g
    return out


def _physical_cores():
    """(physical cores, hardware threads) of this host from /proc/cpuinfo."""
    hw = os.cpu_count() or 1
    try:
        cores, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        return (len(cores) or hw), hw
    except OSError:
        return hw, hw


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path) on a bounded sample of the same workload, protocol of
    SURVEY.md section 8(d): 3 warm-up + 10 timed steps, median; plus a 1-thread number.  torch's intra-op pool at all
    hardware threads is pathological for the many small ops of this path, so the pool size is chosen first by a short
    trial (1 warm-up + 1 timed step at 8 / 32 / 64 threads) and reported."""
    import statistics
    import oracle
    from deltaconv_amd.data import synthetic_batch
    from deltaconv_amd import configs as C
    phys, hw = _physical_cores()
    b = C.make_batch(args.config, args.cpu_clouds, 2, args.points)
    torch.manual_seed(1)
    model = C.build_model(args.config, oracle.models, args.k).train()
    smooth = C.loss_smoothing(args.config)

    def one(batch=b):
        t0 = time.perf_counter()
        model.zero_grad()
        oracle.loss.calc_loss(model(batch), batch.y, smoothing=smooth).backward()
        return time.perf_counter() - t0

    trials = {}
    for threads in sorted({min(hw, t) for t in (8, 32, 64)}):
        torch.set_num_threads(threads)
        one()
        trials[threads] = round(args.cpu_clouds / one(), 3)
    best = max(trials, key=trials.get)
    torch.set_num_threads(best)
    for _ in range(3):
        one()
    times = [one() for _ in range(10)]
    med = statistics.median(times)
    small = C.make_batch(args.config, 2, 3, args.points)
    torch.set_num_threads(1)
    one(small)
    t1 = statistics.median([one(small) for _ in range(3)])
    torch.set_num_threads(best)
    # the SAME inputs as the GPU step: the full batch of the configuration, 1 warm-up + 2 timed steps
    full = None
    if not args.no_cpu_full_batch:
        fb = C.make_batch(args.config, args.batch, 100, args.points)
        one(fb)
        tf = [one(fb) for _ in range(2)]
        full = dict(value=round(args.batch / statistics.median(tf), 3), unit="clouds/s", cores=best, clouds=args.batch,
                    sample=f"the bench batch itself ({args.batch} clouds x {args.points} pts, seed 100), 1 warm-up + 2 timed steps")
    return dict(value=args.cpu_clouds / med, unit="clouds/s", cores=best, kind="port", full_batch=full,
                physical_cores=phys, hardware_threads=hw, one_thread_value=round(2 / t1, 3),
                thread_scaling=("does NOT scale with threads: `cores` is the torch intra-op pool size that measured best, "
                                f"{trials} clouds/s by pool size vs {round(2 / t1, 3)} on ONE thread -- many small ops per step; "
                                "read the value as a roughly single-core restatement, not as a " + str(best) + "-core result"),
                sample=f"oracle/ (torch-CPU restatement of the reference path), {args.cpu_clouds} clouds x "
                       f"{args.points} pts, k={args.k}, fwd+bwd train mode, 3 warm-up + 10 timed steps, median "
                       f"(min {args.cpu_clouds / max(times):.2f} / max {args.cpu_clouds / min(times):.2f} clouds/s) at "
                       f"{best} torch threads (trial clouds/s by threads: {trials}); 1 thread: 2 clouds, median of 3; "
                       f"host: {phys} physical cores / {hw} hardware threads")


def self_launch_command(args, argv, port=None):
    """The command line `python bench.py --gpus N` re-executes itself as when no launcher set WORLD_SIZE: the driver's own
    multi-GPU form, one rank per GPU of this node over RCCL (rendezvous on 127.0.0.1: the container hostname may not
    resolve).  Pure function of its arguments (tests/test_bench_launch.py)."""
    if port is None:
        port = 29500 + os.getpid() % 2000
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves; rank 0 of the child prints the one JSON line
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
        sys.exit(subprocess.call(self_launch_command(args, [a for a in argv if a != "--spawn"]), env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    if world != args.gpus and rank == 0:      # a launcher decides the world size; --gpus only documents it
        print(f"[bench] WORLD_SIZE={world} (launcher) overrides --gpus {args.gpus}", file=sys.stderr)

    import deltaconv_amd as dc  # noqa: F401
    from deltaconv_amd import configs as C
    from deltaconv_amd.utils import calc_loss as _calc_loss
    from deltaconv_amd.dp import FlatGradDataParallel

    cfg = C.CONFIGS[args.config]
    args.batch = per_rank_clouds(args, world)
    smooth = C.loss_smoothing(args.config)
    calc_loss = lambda out, y: _calc_loss(out, y, smoothing=smooth)
    torch.manual_seed(1)
    model = C.build_model(args.config, k=args.k).to(dev).train()
    ddp = FlatGradDataParallel(model, always_reduce=args.force_dist, sync_bn=args.sync_bn and use_dist)
    # the configuration's optimizer (train_modelnet.py:67 / train_shapeseg.py:82), each with its step in ONE launch (deltaconv_amd/optim.py)
    opt = C.build_optimizer(args.config, model.parameters())
    # Inputs resident in HBM before the timed region; every step consumes a different batch.
    batches = [C.make_batch(args.config, args.batch, 100 + rank + 1000 * i, args.points).to(dev)
               for i in range(max(1, args.resident_batches))]
    data = batches[0]
    counter = [0]

    def next_batch():
        counter[0] += 1
        return batches[counter[0] % len(batches)]

    def eager_step():
        b = next_batch()
        ddp.zero_grad()
        loss = calc_loss(ddp(b), b.y)
        loss.backward()
        ddp.reduce_gradients()
        opt.step()
        return loss

    step, launch = eager_step, "eager launches"
    if not args.no_graph:
        # forward + loss + backward (+ SGD update when there is no all-reduce in between) replayed from
        # one captured HIP graph; same kernels, same work per step, one host call.
        from deltaconv_amd.graph_step import GraphedTrainStep
        static = C.make_batch(args.config, args.batch, 99 + rank, args.points).to(dev)
        for _ in range(2):                                   # eager steps first: allocator, optimizer state
            ddp.zero_grad()
            calc_loss(ddp(static), static.y).backward()
            ddp.reduce_gradients()
            opt.step()
        try:
            # one rank: the whole step is one graph.  Data parallel: replay (forward + loss + backward + gradient pack)
            # -> the one all-reduce -> replay (1/world scale + SGD update): three host calls per step
            gstep = GraphedTrainStep(model, calc_loss, static, optimizer=opt, reducer=ddp if use_dist else None)

            def step():
                return gstep(next_batch())
            launch = ("HIP-graph replay" if not use_dist else
                      "one HIP graph with the BatchNorm-statistics and gradient all-reduces captured" if gstep.capture_collectives
                      else "HIP-graph replay x2 around the gradient all-reduce")
        except Exception as e:                               # keep measuring (eagerly) and say so in the output
            print(f"[bench] HIP-graph capture failed, falling back to eager launches: {e!r}", file=sys.stderr)
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    assert torch.isfinite(loss).item(), "loss is not finite"
    # The same step with every dense product on the exact fp32 MFMA chain (option 3; bitwise an fmaf chain): the number the
    # fp32-equivalence claim of the split products travels with.  Not `value`: reported beside it.
    exact_ms = None
    if world == 1 and not use_dist and not args.no_exact_chain and os.environ.get("DC_GEMM_EXACT", "0") in ("", "0"):
        from deltaconv_amd._lib import lib as _lib
        _lib.raw("dc_set_option")(3, 1)
        try:
            estep = eager_step
            if not args.no_graph:
                from deltaconv_amd.graph_step import GraphedTrainStep
                g2 = GraphedTrainStep(model, calc_loss, static, optimizer=opt)
                estep = lambda: g2(next_batch())
            for _ in range(max(2, args.warmup)):
                estep()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                estep()
            torch.cuda.synchronize()
            exact_ms = (time.perf_counter() - t1) / args.steps * 1e3
        except Exception as e:
            print(f"[bench] exact-chain timing failed: {e!r}", file=sys.stderr)
        finally:
            _lib.raw("dc_set_option")(3, 0)
    stamped = None
    if rank == 0 and world == 1 and not use_dist and not args.no_graph and not args.no_in_step_stamps:
        try:
            stamped = in_step_stamps(model, calc_loss, static, opt, args.batch * args.points, args.k)
        except Exception as e:
            print(f"[bench] in-step stamps failed: {e!r}", file=sys.stderr)
    if rank == 0:
        graph, grad, div = model.deltanet_base.build_operators(data)
        roof = apply_roofline(graph, grad, div, cfg["C"])
        graded = [r for r in (stamped or {}).get("div_curl_norm", []) if r["C"] == cfg["C"]]
        if graded:
            # THE graded number: the kernel as it runs inside the replayed training step (its real predecessors, its real
            # operand strides), mean over its instances at C channels; the proxies stay beside it
            ex = sum(r["us"] for r in graded) / len(graded)
            over = (roof.get("dispatch") or {}).get("overhead_us") or 0.0
            us = ex + over
            nbytes = graded[0]["bytes"]
            roof.update(rotating_buffers=dict(achieved=roof["achieved"], frac=roof["frac"], us_per_launch=roof["us_per_launch"]),
                        achieved=round(nbytes / (us * 1e-6) / 1e9, 1), frac=round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        us_per_launch=round(us, 2),
                        execution_only=dict(us=round(ex, 2), achieved=round(nbytes / (ex * 1e-6) / 1e9, 1),
                                            frac=round(nbytes / (ex * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)),
                        in_step_instances=graded, in_step_kernels=stamped,
                        measured=("`achieved` / `frac` / `us_per_launch`: the kernel INSIDE the replayed training step = its execution "
                                  "there (device-clock stamps, first workgroup entry -> last workgroup exit with its stores complete; "
                                  "median of 12 replays, mean over the step's instances at this channel count: `execution_only`) + the "
                                  "dispatch overhead of one launch (`dispatch`: HIP events minus stamps on the same back-to-back "
                                  "launches) -- the duration a rocprofv3 kernel trace of the step shows for it; `rotating_buffers`: HIP "
                                  "events around graph replays on 12 rotating operand sets; `*_l3_resident`: HIP events, one operand set"))
        metric = ("point-clouds/sec fwd+bwd, ModelNet40 1024pt k=20, 1/2/4/8 MI355X" if args.config == "C2" else
                  f"point-clouds/sec fwd+bwd, {cfg['title']} {args.points}pt k={args.k}, 1/2/4/8 MI355X")
        out = {
            "metric": metric,
            "value": args.batch * world * args.steps / dt, "unit": "clouds/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_note": ("fp32 tensors and fp32 accumulation everywhere; the dense per-point products multiply through six bf16 "
                           "partial products of a 3-plane split of each fp32 operand (error against fp64 at or below the exact fp32 "
                           "MFMA chain's: tests/test_gpu_gemm.py; DC_GEMM_EXACT=1 runs the exact chain)"),
            "data": "synthetic (seeded smooth closed surfaces with analytic normals, random-init weights)",
            "config": {"workload": f"{cfg['title']}, {args.points} points, k={args.k}, "
                                   + (f"global batch={args.batch * world} ({args.batch} per GPU, strong scaling), " if args.strong
                                      else f"batch={args.batch} per GPU, ")
                                   + f"fwd+bwd+{'SGD' if cfg['optimizer'] == 'sgd' else 'Adam'} step, train-mode BN/Dropout, "
                                   + ("BatchNorm statistics over the global batch, " if (args.sync_bn and use_dist) else "")
                                   + launch,
                       "name": args.config, "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "roofline": roof,
            "exact_chain_ms_per_step": exact_ms,
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N=1 only (rank 0)
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
