"""Lab: fused row-panel MLP forward (tan_mlp_fwd) vs the four-launch path it replaces -- numerics and timing.  Tool only."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from temporalalignnet_amd import _lib, ops

L = _lib.lib()
dev = "cuda"
torch.manual_seed(0)


def pack(mats, variant):
    """mats: list of [N,K] bf16 tensors -> list of packed flat tensors (one launch)"""
    src = torch.cat([m.reshape(-1) for m in mats])
    dst = torch.empty_like(src)
    ents, off, mx = [], 0, 0
    for m in mats:
        N, K = m.shape
        TN, TK = (512, 16) if N == 512 else (256, 32)
        ents.append((off, off, N, K, TN, TK))
        mx = max(mx, (N // TN) * (K // TK))
        off += N * K
    arr = (_lib.PackEntry * len(ents))(*[_lib.PackEntry(*e) for e in ents])
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    _lib.check(L.tan_pack_weights(ops._ptr(src), ops._ptr(dst), C.c_void_p(tab.data_ptr()), len(ents), mx, ops._stream()), "pack")
    torch.cuda.synchronize()
    outs, off = [], 0
    for m in mats:
        outs.append(dst[off:off + m.numel()])
        off += m.numel()
    return outs


def run_fused(t, variant, store_h=True, next_ln=True, dbg=None):
    d = _lib.MlpDesc()
    d.rows, d.C, d.FF = t["R"], 512, 2048
    d.x_mid = t["x_mid"].data_ptr(); d.ln_g = t["g2"].data_ptr(); d.ln_b = t["b2"].data_ptr()
    d.pw_fc = t["pw"][0].data_ptr(); d.pw_proj = t["pw"][1].data_ptr()
    d.b_fc = t["bfc"].data_ptr(); d.b_proj = t["bpj"].data_ptr()
    d.xn2 = t["f_xn2"].data_ptr(); d.mean2 = t["f_mean2"].data_ptr(); d.rstd2 = t["f_rstd2"].data_ptr()
    d.h_pre = t["f_hpre"].data_ptr()
    d.h_act = t["f_hact"].data_ptr()
    d.x_out = t["f_xout"].data_ptr()
    if next_ln:
        d.nln_g = t["g1"].data_ptr(); d.nln_b = t["b1"].data_ptr(); d.xn_next = t["f_xn1"].data_ptr()
        d.nmean = t["f_mean1"].data_ptr(); d.nrstd = t["f_rstd1"].data_ptr()
    if dbg is not None:
        d.nrstd = dbg.data_ptr()
    d.eps = 1e-5; d.variant = variant
    _lib.check(L.tan_mlp_fwd(C.byref(d), ops._stream()), "tan_mlp_fwd")


def run_unfused(t):
    R = t["R"]
    ops.layernorm_fwd(t["x_mid"], t["g2"], t["b2"], t["u_xn2"], t["u_mean2"], t["u_rstd2"])
    ops.gemm(t["u_xn2"], t["wfc"], t["u_hact"], M=R, N=2048, K=512, bias=t["bfc"], act=ops.ACT_QUICKGELU, aux=t["u_hpre"])
    ops.gemm(t["u_hact"], t["wpj"], t["u_xout"], M=R, N=512, K=2048, bias=t["bpj"], residual=t["x_mid"])
    ops.layernorm_fwd(t["u_xout"], t["g1"], t["b1"], t["u_xn1"], t["u_mean1"], t["u_rstd1"])


def make(R):
    bf = torch.bfloat16
    t = {"R": R}
    t["x_mid"] = (torch.randn(R, 512, device=dev) * 1.5).to(bf)
    t["wfc"] = (torch.randn(2048, 512, device=dev) * 1024 ** -0.5).to(bf)
    t["wpj"] = (torch.randn(512, 2048, device=dev) * 0.03).to(bf)
    t["bfc"] = torch.randn(2048, device=dev) * 0.1
    t["bpj"] = torch.randn(512, device=dev) * 0.1
    for k in ("g1", "g2"):
        t[k] = 1 + 0.1 * torch.randn(512, device=dev)
    for k in ("b1", "b2"):
        t[k] = 0.1 * torch.randn(512, device=dev)
    for pre in ("f_", "u_"):
        for k in ("xn2", "xout", "xn1"):
            t[pre + k] = torch.zeros(R, 512, device=dev, dtype=bf)
        for k in ("hpre", "hact"):
            t[pre + k] = torch.zeros(R, 2048, device=dev, dtype=bf)
        for k in ("mean2", "rstd2", "mean1", "rstd1"):
            t[pre + k] = torch.zeros(R, device=dev)
    t["pw"] = pack([t["wfc"], t["wpj"]], 0)
    return t


def timeit(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def check(R=1024):
    t = make(R)
    run_unfused(t)
    # fp32 reference from the same bf16 inputs
    x = t["x_mid"].float()
    xn2 = torch.nn.functional.layer_norm(x, (512,), t["g2"], t["b2"], 1e-5)
    xn2b = xn2.to(torch.bfloat16).float()
    pre = xn2b @ t["wfc"].float().T + t["bfc"]
    act = pre * torch.sigmoid(1.702 * pre)
    xo = x + act.to(torch.bfloat16).float() @ t["wpj"].float().T + t["bpj"]
    xob = xo.to(torch.bfloat16).float()
    xn1 = torch.nn.functional.layer_norm(xob, (512,), t["g1"], t["b1"], 1e-5)
    ref = {"xn2": xn2, "hpre": pre, "hact": act, "xout": xo, "xn1": xn1, "mean2": x.mean(-1), "rstd2": (x.var(-1, unbiased=False) + 1e-5).rsqrt(),
           "mean1": xob.mean(-1), "rstd1": (xob.var(-1, unbiased=False) + 1e-5).rsqrt()}
    ok = True
    for v in (0,):
        for k in list(t):
            if k.startswith("f_"):
                t[k].zero_()
        run_fused(t, v)
        torch.cuda.synchronize()
        for k, r in ref.items():
            ef = (t["f_" + k].float() - r).abs().max().item()
            eu = (t["u_" + k].float() - r).abs().max().item()
            scale = r.abs().max().item()
            flag = "" if ef <= max(2.5 * eu, 1e-5 * scale) + 1e-6 else "   <-- WORSE THAN UNFUSED"
            if flag:
                ok = False
            print(f"variant {v} {k:6s}: max|fused-ref| {ef:.3e}  max|unfused-ref| {eu:.3e}  (scale {scale:.2f}){flag}")
    print("NUMERICS", "OK" if ok else "MISMATCH")
    return ok


def make_bwd(t):
    """backward inputs for the tensors of make(): dx, packed transposes, gradient accumulators, outputs"""
    R = t["R"]
    bf = torch.bfloat16
    t["dx"] = (torch.randn(R, 512, device=dev) * 0.02).to(bf)
    t["pwt"] = pack([t["wpj"].T.contiguous(), t["wfc"].T.contiguous()], 0)
    t["dh"] = torch.zeros(R, 2048, device=dev, dtype=bf)
    t["dx2"] = torch.zeros(R, 512, device=dev, dtype=bf)
    t["dxn"] = torch.zeros(R, 512, device=dev, dtype=bf)
    for k, n in (("g_b_fc", 2048), ("g_ln_g", 512), ("g_ln_b", 512), ("g_b_out", 512)):
        t[k] = torch.zeros(n, device=dev)
    t["ln_ws"] = torch.zeros(4 * 1024 * 512, device=dev)
    return t


def run_bwd(t):
    d = _lib.MlpBwdDesc()
    d.rows, d.C, d.FF = t["R"], 512, 2048
    d.dx, d.h_pre, d.x_mid = t["dx"].data_ptr(), t["f_hpre"].data_ptr(), t["x_mid"].data_ptr()
    d.mean2, d.rstd2, d.ln_g = t["f_mean2"].data_ptr(), t["f_rstd2"].data_ptr(), t["g2"].data_ptr()
    d.pwt_proj, d.pwt_fc = t["pwt"][0].data_ptr(), t["pwt"][1].data_ptr()
    d.dh, d.dx2 = t["dh"].data_ptr(), t["dx2"].data_ptr()
    d.g_b_fc, d.g_ln_g, d.g_ln_b, d.g_b_out = (t[k].data_ptr() for k in ("g_b_fc", "g_ln_g", "g_ln_b", "g_b_out"))
    _lib.check(L.tan_mlp_bwd(C.byref(d), ops._stream()), "tan_mlp_bwd")


def bwd_main():
    for R in (8192, 10240):
        ts = [make_bwd(make(R)) for _ in range(6)]
        for t in ts:
            run_fused(t, 0)            # h_pre, mean2, rstd2 of the forward
        fl = 2.0 * R * 512 * 2048 * 2
        it = [0]

        def g():
            run_bwd(ts[it[0] % 6]); it[0] += 1
        tb = timeit(g, reps=48)
        print(f"R={R} cold: fused backward                  {tb:7.1f} us   {fl / tb / 1e6:6.0f} TF/s", flush=True)
        t = ts[0]
        tb = timeit(lambda: run_bwd(t))
        print(f"R={R} warm: fused backward                  {tb:7.1f} us   {fl / tb / 1e6:6.0f} TF/s", flush=True)
        if R == 8192 and os.environ.get("TAN_PANEL_LAB_CLOCKS"):
            dbg = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
            assert L.tan_panel_lab_set_dbg(C.c_void_p(dbg.data_ptr())) == 0
            for i in range(7):
                run_bwd(ts[i % 6])
            torch.cuda.synchronize()
            st = dbg.cpu().numpy().reshape(8, 64)
            t0 = st[:, 0].min()
            names = ["fc0", "pe0"] + [f"{k}{c}" for c in range(1, 8) for k in ("fc", "pe")] + ["pj7"]
            for w in (0, 4):
                row = st[w]
                print(f" wave {w}: total {int(row[35] - row[0])}  " + " ".join(
                    f"{n}:{int(row[1 + 2 * k] - t0)}+{int(row[2 + 2 * k] - row[1 + 2 * k])}" for k, n in enumerate(names)))
        del ts, t
        torch.cuda.empty_cache()


def phase_clocks(variant, R=8192):
    """MODE & 64: per-wave shader-clock stamps at the phase boundaries of workgroup 0 (cold buffers)"""
    ts = [make(R) for _ in range(6)]
    dbg = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
    assert L.tan_panel_lab_set_dbg(C.c_void_p(dbg.data_ptr())) == 0
    for i in range(7):
        run_fused(ts[i % 6], variant, True, True)
    torch.cuda.synchronize()
    return dbg.cpu().numpy().reshape(8, 64)


def print_clocks(variant):
    st = phase_clocks(variant)
    t0 = st[:, 0].min()
    names = ["fc0", "pe0"] + [f"{k}{c}" for c in range(1, 8) for k in ("fc", "pe")] + ["pj7"]
    print(f"variant {variant}: shader clocks per phase (workgroup 0); start offset / duration per wave")
    for w in range(8):
        row = st[w]
        dur = [int(row[2 + 2 * k] - row[1 + 2 * k]) for k in range(17)]
        beg = [int(row[1 + 2 * k] - t0) for k in range(17)]
        print(f" wave {w}: total {int(row[35] - row[0])}  " + " ".join(f"{n}:{b}+{d}" for n, b, d in zip(names, beg, dur)))



if __name__ == "__main__":
    if os.environ.get("LAB_BWD"):
        bwd_main()
        sys.exit(0)
    if os.environ.get("LAB_CLOCKS"):
        for v in (int(x) for x in os.environ["LAB_CLOCKS"].split(",")):
            print_clocks(v)
        sys.exit(0)
    check(1024)
    variants = tuple(int(v) for v in os.environ.get("LAB_VARIANTS", "0,16,32,8,5,2,15").split(","))
    # COLD: six layers' worth of distinct weights and output buffers cycled (as in a stack: nothing is L2 / Infinity-Cache warm)
    for R in (8192, 10240):
        ts = [make(R) for _ in range(6)]
        fl = 2.0 * R * 512 * 2048 * 2
        it = [0]

        def cyc(f):
            def g():
                f(ts[it[0] % 6]); it[0] += 1
            return g
        tu = timeit(cyc(run_unfused), reps=48)
        print(f"R={R} cold: unfused (LN, fc, proj, LN)       {tu:7.1f} us   {fl / tu / 1e6:6.0f} TF/s", flush=True)
        for v in variants:
            tf = timeit(cyc(lambda t: run_fused(t, v, True, True)), reps=48)
            print(f"R={R} cold: fused v{v:<2d}                        {tf:7.1f} us   {fl / tf / 1e6:6.0f} TF/s", flush=True)
        if os.environ.get("LAB_E3"):
            # all cold, but the packed weights are touched (read once: Infinity Cache / one L2) right before the launch
            wsum = [torch.zeros(1, device=dev) for _ in range(6)]

            def touch(i):
                for w in ts[i % 6]["pw"]:
                    wsum[i % 6] += w.view(torch.int16)[::64].sum()       # one element per 128-byte line
            it4 = [0]

            def g_touch_only():
                touch(it4[0]); it4[0] += 1

            def g_both():
                touch(it4[0]); run_fused(ts[it4[0] % 6], 0, True, True); it4[0] += 1
            t_touch = timeit(g_touch_only, reps=48)
            t_both = timeit(g_both, reps=48)
            print(f"R={R} cold + weights touched first: touch {t_touch:6.1f} us, touch + fused v0 {t_both:6.1f} us -> fused alone ~{t_both - t_touch:6.1f} us", flush=True)
        if os.environ.get("LAB_E4"):
            # the UNFUSED quartet: which coldness costs its 20 us (weights: 4 MB, activations / outputs: ~150 MB)?
            def mixu(i, what):
                t = dict(ts[0])
                src = ts[i % 6]
                keys = ("wfc", "wpj") if what == "w" else [k for k in src if k not in ("pw", "wfc", "wpj", "R")]
                for k in keys:
                    t[k] = src[k]
                return t
            for what in ("w", "o"):
                mixed = [mixu(i, what) for i in range(6)]
                it5 = [0]

                def gu():
                    run_unfused(mixed[it5[0] % 6]); it5[0] += 1
                tf = timeit(gu, reps=48)
                print(f"R={R} cold-{what} only: unfused                  {tf:7.1f} us", flush=True)
            for nm, (M_, N_, K_) in {"fc": (R, 2048, 512), "proj": (R, 512, 2048)}.items():
                for what in ("warm", "w", "o", "all"):
                    it6 = [0]

                    def gg():
                        i = it6[0] % 6; it6[0] += 1
                        tw = ts[i if what in ("w", "all") else 0]
                        to = ts[i if what in ("o", "all") else 0]
                        if nm == "fc":
                            ops.gemm(to["u_xn2"], tw["wfc"], to["u_hact"], M=R, N=2048, K=512, bias=tw["bfc"], act=ops.ACT_QUICKGELU, aux=to["u_hpre"])
                        else:
                            ops.gemm(to["u_hact"], tw["wpj"], to["u_xout"], M=R, N=512, K=2048, bias=tw["bpj"], residual=to["x_mid"])
                    tf = timeit(gg, reps=48)
                    print(f"R={R} gemm {nm:4s} cold={what:4s}  {tf:7.1f} us", flush=True)
        if os.environ.get("LAB_E2"):
            # which coldness matters: (a) cycling weights only, (b) cycling activations / outputs only
            def mix(i, what):
                t = dict(ts[0])
                src = ts[i % 6]
                keys = ("pw", "wfc", "wpj") if what == "w" else [k for k in src if k not in ("pw", "wfc", "wpj", "R")]
                for k in keys:
                    t[k] = src[k]
                return t
            for what in ("w", "o"):
                mixed = [mix(i, what) for i in range(6)]
                it3 = [0]

                def g():
                    run_fused(mixed[it3[0] % 6], 0, True, True); it3[0] += 1
                tf = timeit(g, reps=48)
                print(f"R={R} cold-{what} only: fused v0                 {tf:7.1f} us", flush=True)
        t = ts[0]
        tu = timeit(lambda: run_unfused(t))
        print(f"R={R} warm: unfused (LN, fc, proj, LN)       {tu:7.1f} us   {fl / tu / 1e6:6.0f} TF/s", flush=True)
        for v in variants:
            tf = timeit(lambda: run_fused(t, v, True, True))
            print(f"R={R} warm: fused v{v:<2d}                        {tf:7.1f} us   {fl / tf / 1e6:6.0f} TF/s", flush=True)
        del ts, t
        torch.cuda.empty_cache()
    # two panels' worth of work on two streams at once (video + joint stack sizes), cold
    tas, tbs = [make(8192) for _ in range(6)], [make(10240) for _ in range(6)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    it2 = [0]

    def both(fa, fb):
        i = it2[0] % 6; it2[0] += 1
        with torch.cuda.stream(s1):
            fa(tas[i])
        with torch.cuda.stream(s2):
            fb(tbs[i])

    def time_both(fa, fb, reps=48):
        both(fa, fb); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s1.wait_event(e0); s2.wait_event(e0)
        for _ in range(reps):
            both(fa, fb)
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    fl2 = 2.0 * (8192 + 10240) * 512 * 2048 * 2
    tu = time_both(run_unfused, run_unfused)
    print(f"two streams 8192+10240 cold: unfused  {tu:7.1f} us {fl2 / tu / 1e6:6.0f} TF/s")
    for v in variants[:4]:
        tf = time_both(lambda t: run_fused(t, v), lambda t: run_fused(t, v))
        print(f"two streams 8192+10240 cold: fused v{v:<2d} {tf:7.1f} us {fl2 / tf / 1e6:6.0f} TF/s")

