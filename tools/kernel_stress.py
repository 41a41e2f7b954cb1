"""Round-6 stress: does ONE kernel, launched over and over on identical inputs, return identical bits while a second process does the same on the same GPU?

    python tools/kernel_stress.py [procs] [iters] [KEY=VALUE ...]

tools/diag_r06.py found that every 2-rank deviation of the BC gradients starts at the dx output of a vpt_ln_bwd_kernel launch -- one whole row, inputs
clean.  This tool takes the process group, the trainer and the model away: each process builds fixed inputs, computes every candidate kernel's output once
(alone), then -- all processes started together by a file barrier -- repeats the launches `iters` times and compares each output with its first one bit for
bit on the device.  A mismatching output is kept (the first few) and described: which rows, how large, scaled or shifted.
Candidates: layernorm_backward (M = 10 and 70, D = 1024 / 2048, with and without dx_add / relu_in), layernorm forward, the split-K linear + epilogue that
produces layernorm_backward's input in the trainer, gate_cast, column_sum."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vpt_amd  # noqa: E402,F401


def _barrier(d, tag, rank, world):
    open(os.path.join(d, f"bar_{tag}_{rank}"), "w").close()
    while not all(os.path.exists(os.path.join(d, f"bar_{tag}_{r}")) for r in range(world)):
        pass


def describe(name, ref, got):
    r, g = ref.double().reshape(ref.shape[0], -1), got.double().reshape(ref.shape[0], -1)
    bad_rows = ((r != g).sum(1) > 0).nonzero().flatten().tolist()
    out = [f"{name}: {int((r != g).sum())} of {r.numel()} elements differ, rows {bad_rows[:8]}"]
    for row in bad_rows[:2]:
        a, b = r[row], g[row]
        rel = float((a - b).norm() / a.norm().clamp_min(1e-30))
        # is the wrong row  alpha * right + beta ?  (a wrong scale / shift scalar of the row)
        A = torch.stack([a, torch.ones_like(a)], 1)
        sol = torch.linalg.lstsq(A, b.unsqueeze(1)).solution.flatten()
        resid = float((A @ sol.unsqueeze(1) - b.unsqueeze(1)).norm() / b.norm().clamp_min(1e-30))
        out.append(f"    row {row}: rel {rel:.3e}; best fit wrong = {float(sol[0]):.8f} * right + {float(sol[1]):.3e} (residual {resid:.2e}); first elements right {a[:4].tolist()} wrong {b[:4].tolist()}")
    return "\n".join(out)


def patch_layernorm_backward(ops, hsaco):
    """STRESS_LN_HSACO=<code object>: ops.layernorm_backward launches vpt_ln_bwd_kernel / vpt_ln_bwd_finish_kernel from THAT code object (hipModuleLaunchKernel)
    instead of the library's -- everything else of the stress unchanged.  tools/ubench/pk_hazard/make_variants.py builds the SLP-packed kernel and variants of
    its assembly with s_nop inserted: which of them still fail here bisects the fault at the ISA level."""
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    mod = ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), hsaco.encode()) == 0, hsaco
    fns = {}
    for nd4 in (4, 8, 12, 16):
        f = ctypes.c_void_p()
        assert hip.hipModuleGetFunction(ctypes.byref(f), mod, f"_Z17vpt_ln_bwd_kernelILi{nd4}EEv12VptLnBwdArgs".encode()) == 0
        fns[nd4] = f
    fin = ctypes.c_void_p()
    assert hip.hipModuleGetFunction(ctypes.byref(fin), mod, b"_Z24vpt_ln_bwd_finish_kernelPKfiiPfS1_") == 0

    debug = os.environ.get("STRESS_LN_DEBUG") == "1"      # debug.hsaco: the kernel also writes every lane's partial (s1, s2) and the reduced (s1, s2) of every row

    class A(ctypes.Structure):
        _fields_ = [(n, ctypes.c_void_p) for n in ("x", "gain", "dy", "dx_add", "dx", "dgain", "dbias", "partials")] + [(n, ctypes.c_int) for n in ("M", "D", "relu_in")] + \
                   ([("pad", ctypes.c_int), ("debug", ctypes.c_void_p)] if debug else [])

    class F(ctypes.Structure):
        _fields_ = [("partials", ctypes.c_void_p), ("nblocks", ctypes.c_int), ("D", ctypes.c_int), ("dgain", ctypes.c_void_p), ("dbias", ctypes.c_void_p)]

    def launch(fn, grid, st):
        size = ctypes.c_size_t(ctypes.sizeof(st))
        extra = (ctypes.c_void_p * 5)(1, ctypes.cast(ctypes.pointer(st), ctypes.c_void_p), 2, ctypes.cast(ctypes.pointer(size), ctypes.c_void_p), 3)
        assert hip.hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None, extra) == 0

    def layernorm_backward(x, gain, dy, dgain, dbias, relu_in=False, dx_add=None):
        m, d = x.shape
        dx = torch.empty_like(x)
        g = (m + 31) // 32
        part = torch.empty(4 * g * 2 * d, dtype=torch.float32, device=x.device)
        nd4 = ((d >> 2) + 63) >> 6
        key = 4 if nd4 <= 4 else (8 if nd4 <= 8 else (12 if nd4 <= 12 else 16))
        st = A(x.data_ptr(), gain.data_ptr(), dy.data_ptr(), dx_add.data_ptr() if dx_add is not None else None, dx.data_ptr(), dgain.data_ptr(),
               dbias.data_ptr(), part.data_ptr(), m, d, 1 if relu_in else 0)
        if debug:
            ops._ln_debug_last = torch.zeros(m, 64, 4, dtype=torch.float32, device=x.device)
            st.debug = ops._ln_debug_last.data_ptr()
        launch(fns[key], g, st)
        launch(fin, (2 * d + 15) // 16, F(part.data_ptr(), 4 * g, d, dgain.data_ptr(), dbias.data_ptr()))
        return dx

    ops.layernorm_backward = layernorm_backward
    ops._pk_hazard_keep = (hip, mod)


def forensic(inputs, ref, got):
    """A wrong row of dx is right + beta.  dx = rstd (dy g - s1 - xhat s2): beta = rstd (s1_right - s1_wrong).  Which part of s1 = (1/D) sum_i dy_i g_i is missing?
    Candidates: the contribution of ONE vector instruction (element slot (q, k) of every lane: i = 4 (lane + 64 q) + k), of a whole float4 slot q, or of one side of a
    butterfly step of the wave reduction (the lanes with bit b set / clear).  Reports the best match."""
    x, dy, gain, relu_in = (t.double().cpu() if torch.is_tensor(t) else t for t in inputs)
    r, g = ref.double(), got.double()
    rows = ((r != g).sum(1) > 0).nonzero().flatten().tolist()
    out = []
    d_ = x.shape[1]
    nd4 = ((d_ >> 2) + 63) >> 6
    for row in rows[:3]:
        xr = x[row].clamp_min(0) if relu_in else x[row]
        mean = xr.mean()
        rstd = 1.0 / torch.sqrt(((xr - mean) ** 2).mean() + 1e-5)
        dg = dy[row] * gain
        beta = float((g[row] - r[row]).mean())
        want = beta / float(rstd) * d_                    # the missing part of sum_i dy_i g_i  (s1_right - s1_wrong) * D
        idx = torch.arange(d_)
        lane, q, k = (idx // 4) % 64, (idx // 4) // 64, idx % 4
        cands = {}
        for qq in range(nd4):
            cands[f"float4 slot q={qq} (all lanes)"] = float(dg[q == qq].sum())
            for kk in range(4):
                cands[f"element slot (q={qq}, k={kk}) of every lane"] = float(dg[(q == qq) & (k == kk)].sum())
            for kk in (0, 2):
                cands[f"element pair (q={qq}, k={kk},{kk + 1}) of every lane"] = float(dg[(q == qq) & ((k == kk) | (k == kk + 1))].sum())
        for b in range(6):
            cands[f"lanes with bit {b} set"] = float(dg[((lane >> b) & 1) == 1].sum())
            cands[f"lanes with bit {b} clear"] = float(dg[((lane >> b) & 1) == 0].sum())
        cands["everything (s1 = 0)"] = float(dg.sum())
        best = sorted([(abs(v - want), n, v) for n, v in cands.items()] + [(abs(-v - want), "MINUS " + n, -v) for n, v in cands.items()])[:3]
        out.append(f"    forensic row {row}: shift {beta:.6e}, i.e. sum dy g is off by {want:.6e}; closest candidates: " +
                   " | ".join(f"{n}: {v:.6e} (|diff| {e:.1e})" for e, n, v in best))
    return "\n".join(out)


def debug_report(inputs, r_, g_):
    """The instrumented kernel's buffer [M][64 lanes][partial s1, partial s2, reduced s1, reduced s2]: which lanes' PARTIAL sums are wrong, and is the error of
    each such lane one term dy_i g_i of that lane (missing: right - term, or added twice: right + term) -- the same element slot (q, k) for all of them?"""
    out = []
    for row in ((r_ != g_).flatten(1).sum(1) > 0).nonzero().flatten().tolist()[:3]:
        parts = []
        for c, nm in enumerate(("partial s1", "partial s2", "reduced s1", "reduced s2")):
            bad_l = (r_[row, :, c] != g_[row, :, c]).nonzero().flatten().tolist()
            if bad_l:
                parts.append(f"{nm}: {len(bad_l)} lanes differ {bad_l[0]}..{bad_l[-1]} (lane {bad_l[0]}: right {r_[row, bad_l[0], c]:.9e} wrong {g_[row, bad_l[0], c]:.9e})")
        out.append(f"    debug row {row}: " + (" | ".join(parts) if parts else "equal"))
        bad_l = (r_[row, :, 0] != g_[row, :, 0]).nonzero().flatten()
        if inputs is not None and len(bad_l):
            x, dy, gain, relu_in = (t.double().cpu() if torch.is_tensor(t) else t for t in inputs)
            d_ = x.shape[1]
            nd4 = ((d_ >> 2) + 63) >> 6
            dg = (dy[row] * gain).view(-1, 4)                      # [float4 index i = lane + 64 q][k]
            diff = g_[row, bad_l, 0] - r_[row, bad_l, 0]
            best = []
            for q in range(nd4):
                for k in range(4):
                    term = dg[bad_l + 64 * q, k]
                    for sign, what in ((-1.0, "missing"), (1.0, "added twice")):
                        best.append((float((diff - sign * term).abs().max() / diff.abs().max().clamp_min(1e-300)), f"term (q={q}, k={k}) {what}"))
                for sign, what in ((-1.0, "missing"), (1.0, "added twice")):
                    best.append((float((diff - sign * dg[bad_l + 64 * q].sum(1)).abs().max() / diff.abs().max().clamp_min(1e-300)), f"whole float4 q={q} {what}"))
            best.sort()
            out.append(f"      partial-s1 error of those lanes vs one term of the lane: best {best[0][1]} (relative misfit {best[0][0]:.2e}), then {best[1][1]} ({best[1][0]:.2e}), {best[2][1]} ({best[2][0]:.2e})")
    return "\n".join(out)


def worker(rank, world, d, iters):
    from vpt_amd import ops, packing
    if os.environ.get("STRESS_LN_HSACO"):
        patch_layernorm_backward(ops, os.environ["STRESS_LN_HSACO"])
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    cases = {}

    def ln_case(m, dd, relu_in, add):
        x = torch.randn(m, dd, generator=g).to(dev)
        dy = (torch.randn(m, dd, generator=g) * 1e-3).to(dev)
        gain = (1 + 0.1 * torch.randn(dd, generator=g)).to(dev)
        dxa = torch.randn(m, dd, generator=g).to(dev) * 1e-3 if add else None

        def run():
            dg, db = torch.zeros(dd, device=dev), torch.zeros(dd, device=dev)
            dx = ops.layernorm_backward(x, gain, dy, dg, db, relu_in=relu_in, dx_add=dxa)
            if getattr(ops, "_ln_debug_last", None) is not None:
                return dx, dg, db, ops._ln_debug_last
            return dx, dg, db
        run.inputs = (x, dy, gain, relu_in)
        return run

    for m, dd in ((10, 1024), (70, 1024), (10, 2048)):
        cases[f"ln_bwd M={m} D={dd}"] = ln_case(m, dd, False, False)
        cases[f"ln_bwd M={m} D={dd} relu_in dx_add"] = ln_case(m, dd, True, True)
    xf = torch.randn(10, 1024, generator=g).to(dev)
    gf, bf = (1 + 0.1 * torch.randn(1024, generator=g)).to(dev), (0.1 * torch.randn(1024, generator=g)).to(dev)
    cases["ln_fwd M=10 D=1024"] = lambda: ops.layernorm(xf, gf, bf, out_f32=True)
    w = (torch.randn(1024, 4096, generator=g) * 0.02).to(dev)
    wpk = ops.pack_linear(w.contiguous())
    a16 = (torch.randn(10, 4096, generator=g) * 1e-2).to(torch.bfloat16).to(dev)
    res = torch.randn(10, 1024, generator=g).to(dev)
    cases["linear split-K M=10 4096->1024 (+res)"] = lambda: ops.linear(a16, wpk, 1024, res=res)[:1]
    cases["gate_cast M=10"] = lambda: (ops.gate_cast(xf, 1024),)

    def chain():      # the trainer's sequence: split-K linear -> layernorm_backward on its output
        o32, _ = ops.linear(a16, wpk, 1024, res=res)
        dg, db = torch.zeros(1024, device=dev), torch.zeros(1024, device=dev)
        return ops.layernorm_backward(xf, gf, o32, dg, db), o32
    cases["chain linear -> ln_bwd"] = chain

    # ---- whole paths (fewer launches each): the inference forward (throughput tiling), the acting step (latency tiling: GEMV kernels), the IDM
    # (temporal conv), one BC gradient computation (every backward kernel) ----
    heavy = {}
    if os.environ.get("STRESS_PATHS", "1") == "1":
        from vpt_amd import configs
        from vpt_amd.lib.policy import MinecraftAgentPolicy, InverseActionPolicy
        from vpt_amd.lib.types import minecraft_action_space, idm_action_space
        from vpt_amd.training import BCTrainer
        prec, width = os.environ.get("STRESS_PRECISION", "bf16"), os.environ.get("STRESS_WIDTH", "1x")     # the fp16 library is a separate binary
        pol = MinecraftAgentPolicy(minecraft_action_space(), configs.policy_kwargs_for(width), dict(temperature=2.0), precision=prec)
        configs.randomize_(pol, 0)
        pol = pol.to(dev)
        img = torch.randint(0, 256, (2, 8, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev)
        first = torch.zeros(2, 8, dtype=torch.bool, device=dev)

        def fwd():
            with torch.no_grad():
                (pd, v, _), st = pol({"img": img}, first, pol.initial_state(2))
            return pd["buttons"], pd["camera"], v, st[-1][1][0]

        def act():
            with torch.no_grad():
                (pd, v, _), st = pol({"img": img[:1, :1]}, first[:1, :1], pol.initial_state(1))
            return pd["buttons"], v, st[0][1][1]
        heavy[f"policy forward {width} {prec} B=2 T=8"] = fwd
        heavy[f"acting step {width} {prec} B=1 T=1"] = act
        idm = InverseActionPolicy(idm_action_space(), pi_head_kwargs=dict(temperature=2.0), idm_net_kwargs=configs.idm_kwargs_for("tiny"), precision=prec)
        configs.randomize_(idm, 1)
        idm = idm.to(dev)
        vid = torch.randint(0, 256, (1, 16, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev)

        def idm_fwd():
            with torch.no_grad():
                (pd, _, _), _ = idm({"img": vid}, torch.zeros(1, 16, dtype=torch.bool, device=dev), idm.initial_state(1))
            return pd["buttons"], pd["camera"]
        heavy[f"IDM tiny {prec} T=16"] = idm_fwd
        tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
        ab, ac = torch.randint(0, 8641, (2, 8), generator=g).to(dev), torch.randint(0, 121, (2, 8), generator=g).to(dev)

        def bc():
            loss, grads, _ = tr.loss_and_grads(img, first, pol.initial_state(2), ab, ac)
            return [loss] + [grads[k] for k in sorted(grads)]
        heavy[f"BC gradients {width} {prec} B=2 T=8 (all tensors)"] = bc
    for k, f in heavy.items():
        try:
            f()
            cases[k] = f
        except Exception as e:      # a path this tool drives wrongly is reported, not fatal
            print(f"  rank {rank}: case {k} skipped: {type(e).__name__}: {e}"[:300], flush=True)
    side = torch.cuda.Stream() if os.environ.get("STRESS_SIDE_STREAM") == "1" else None
    noise = torch.randn(1 << 24, device=dev) if side is not None else None

    refs = {k: [t.clone() for t in f() if t is not None] for k, f in cases.items()}
    torch.cuda.synchronize()
    _barrier(d, "start", rank, world)
    lines, t0 = [], time.time()
    for name, f in cases.items():
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        last_bad = [r.clone() for r in refs[name]]            # per output: the most recent mismatching result (device-side select, no synchronisation)
        n_it = iters if name not in heavy else max(20, iters // (40 if name.startswith("BC") else 10))
        for it in range(n_it):
            if side is not None:        # a busy neighbour INSIDE the process: a long element-wise kernel on a second stream under every launch
                with torch.cuda.stream(side):
                    noise.mul_(1.0000001).add_(1e-9)
            outs = [t for t in f() if t is not None]
            for j, (o, r) in enumerate(zip(outs, refs[name])):
                ne = (o.view(torch.int16) != r.view(torch.int16)).any() if o.dtype in (torch.bfloat16, torch.float16) else (o != r).any()
                bad += ne.to(torch.int64)
                last_bad[j] = torch.where(ne, o, last_bad[j])
        torch.cuda.synchronize()
        lines.append(f"  rank {rank} {name}: {int(bad.item())} mismatching outputs in {n_it} launches")
        for j, o in enumerate(last_bad):
            if not torch.equal(o, refs[name][j]) and o.dim() == 3 and tuple(o.shape[1:]) == (64, 4):
                lines.append(debug_report(getattr(f, "inputs", None), refs[name][j].double().cpu(), o.double().cpu()))
            elif not torch.equal(o, refs[name][j]) and o.dim() >= 2:
                lines.append("  " + describe(f"output {j}", refs[name][j].float().cpu(), o.float().cpu()))
                if j == 0 and hasattr(f, "inputs"):
                    lines.append(forensic(f.inputs, refs[name][j].float().cpu(), o.float().cpu()))
            elif not torch.equal(o, refs[name][j]):
                lines.append(f"    output {j} (shape {tuple(o.shape)}) differs")
    lines.append(f"  rank {rank}: {time.time() - t0:.1f} s")
    with open(os.path.join(d, f"out{rank}.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")


def main():
    import torch.multiprocessing as mp
    nums = [a for a in sys.argv[1:] if "=" not in a]
    procs = int(nums[0]) if nums else 2
    iters = int(nums[1]) if len(nums) > 1 else 2000
    os.environ.update(dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a))
    print(f"=== kernel_stress procs {procs} iters {iters} env {[a for a in sys.argv[1:] if '=' in a]}", flush=True)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(procs, d, iters), nprocs=procs, join=True)
        for r in range(procs):
            print(open(os.path.join(d, f"out{r}.txt")).read().rstrip(), flush=True)


if __name__ == "__main__":
    main()
