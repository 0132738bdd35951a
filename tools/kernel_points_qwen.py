import sys, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
for B, V, sc in ((1024, 151936, False), (1024, 151936, True), (768, 151936, False), (2048, 151936, False), (3072, 151936, False), (4096, 32000, True)):
    r = bench.kernel_point(dev, B, V, scores=sc, iters=50)
    print(json.dumps({"B": B, "V": V, "scores": sc, "us": r["launch_us"], "frac_alg": r["frac"], "frac_traffic": r.get("frac_traffic")}), flush=True)
