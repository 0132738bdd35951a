"""Persistent few-row decode layers against the five-launch layer chain: time per step of n_layers at 7B (or 13B) widths, both
captured in a HIP graph (what the engine replays).  python tools/probes/lost_kernels/persistent_probe.py [--layers 32] [--rows 2] [--model 7b]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))                       # lost_ops.py, the lab tests
import torch
import test_persistent_layers_gpu as T
import lost_ops as ops                   # lab entries + the product's ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--rows", type=int, default=2)
    ap.add_argument("--model", default="7b")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--ctx", type=int, default=650)
    ap.add_argument("--timeline", action="store_true", help="per-phase timeline of one launch (mean over workgroups and layers >= 1)")
    a = ap.parse_args()
    T.DT = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    d, H, F = (4096, 32, 11008) if a.model == "7b" else (5120, 40, 13824)
    M, eps = a.rows, 1e-5
    lens = [a.ctx, 75, a.ctx - 100, 60][:M]
    plens = [a.ctx - 39, 36, a.ctx - 150, 0][:M]
    L, rows, pos, cpos, slot, x, cs = T._setup(ops, M, d, H, F, a.layers, lens, plens, seed=1)
    out = dict(model=a.model, layers=a.layers, rows=M, dtype=a.dtype, ctx=a.ctx,
               max_rows=ops.decode_layers_max_rows(d, H, H, F, 128, a.layers, T.DT))
    desc = ops.layer_descriptors(L, T.DEV)
    ws = ops.decode_layers_workspace(M, d, H, F, 128, T.DEV, T.DT)
    l0 = L[0]

    def persistent():
        return ops.decode_layers(desc, len(L), x, pos, cpos, slot, cs, rows, H, F, 128, eps, l0["k_own"].stride(0), l0["k_own"].shape[2],
                                 l0["k_pre"].stride(0), l0["k_pre"].shape[2], False, ws)

    def chain():
        return T._five_launch(ops, L, x, pos, cpos, slot, cs, rows, H, F, eps)

    for name, fn in (("five_launch", chain), ("persistent", persistent)):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                r = fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if name == "persistent":
            out["status_after_warmup"] = ops.decode_layers_status(ws)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            r = fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        out[name + "_us_per_layer"] = round(dt / a.layers * 1e6, 2)
        out[name + "_ms"] = round(dt * 1e3, 3)
        out[name + "_resid_absmax"] = float(r[0].float().abs().max())
    out["status"] = ops.decode_layers_status(ws)
    if a.timeline:
        G = min(torch.cuda.get_device_properties(0).multi_processor_count, d // 16)
        big = torch.zeros(ws.numel() + G * a.layers * 256, dtype=torch.uint8, device=T.DEV)
        for _ in range(3):
            ops.decode_layers(desc, len(L), x, pos, cpos, slot, cs, rows, H, F, 128, eps, l0["k_own"].stride(0), l0["k_own"].shape[2],
                              l0["k_pre"].stride(0), l0["k_pre"].shape[2], False, big)
        torch.cuda.synchronize()
        tl = big[ws.numel():].view(torch.int64).view(G, a.layers, 32).cpu().double()
        names = ["qkv_wait", "qkv_go", "o_wait", "o_go", "gu_wait", "gu_go", "down_wait", "down_go", "att_start", "att_qkv_ready", "att_end",
                 "G1_done", "G3_done", "G4_done", "G5_done", "G6_done", "G1_start", "G1_pass1", "G1_tags_ok", "G1_gbar1", "G1_normed", "G1_retries",
                 "down_fin_wait", "down_fin_go", "down_published"]
        rel = (tl[:, 1:, :] - tl[:, 1:, 0:1]) / 100.0                       # us relative to the layer's qkv_wait stamp, layers >= 1
        att = rel[: H * 8] if H * 8 <= G else rel
        out["timeline_us_mean"] = {n: round(float((att if n.startswith("att") else rel)[:, :, i].mean()), 2) for i, n in enumerate(names)}
        out["timeline_us_max_over_cus"] = {n: round(float((att if n.startswith("att") else rel)[:, :, i].max(0).values.mean()), 2) for i, n in enumerate(names)}
        out["timeline_us_mean"]["G1_retries"] = round(float(tl[:, 1:, 21].mean()), 2)
        # absolute skew: when do the workgroups publish the layer's last output, relative to the first one to do so
        pub = tl[:, 1:-1, 24]
        out["down_published_spread_us"] = round(float((pub.max(0).values - pub.min(0).values).mean()) / 100.0, 2)
        out["down_published_to_next_qkv_go_us"] = round(float((tl[:, 2:, 1].max(0).values - pub.max(0).values).mean()) / 100.0, 2)
        out["layer_period_us"] = round(float((tl[:, 2:, 0] - tl[:, 1:-1, 0]).mean()) / 100.0, 2) if a.layers > 2 else None
    out["ratio"] = round(out["persistent_ms"] / out["five_launch_ms"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
