"""HBM bytes of the decode attention kernels at the headline batch (1,536 rows: 128 images x 6 questions x 2 branches) from the PMC counters:
`rocprofv3 --pmc FETCH_SIZE --output-format csv -d D -o k -- python tools/attn_pmc.py`, then `attn_pmc.py analyse <counter_collection.csv> FETCH_SIZE`
(one counter per pass, as the guide prescribes).  The engine runs WITHOUT graph capture here so that every launch is a dispatch of its own."""
import csv, os, statistics, sys, collections, re
if sys.argv[1:2] == ["analyse"]:
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        if r["Counter_Name"] == sys.argv[3] and ("decode_attn" in r["Kernel_Name"] or "rope_kv" in r["Kernel_Name"] or "rmsnorm" in r["Kernel_Name"]):
            n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:60]
            rows[(n, r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (n, g), v in sorted(rows.items(), key=lambda kv: -statistics.mean(kv[1])):
        print(f"{sys.argv[3]:11s} mean {statistics.mean(v):12.1f} KB  (min {min(v):12.1f}, max {max(v):12.1f})  n={len(v):5d}  grid={g:>9s}  {n}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import pope_prompts
from llava_align_amd.engine import VddLlavaEngine
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=False, lm_head_gain=4.0)
ids, host_imgs = pope_prompts(128, seed=1234, vocab=eng.cfg.lm.vocab, image=eng.cfg.vision.image)
on_dev = {}
imgs = [on_dev.setdefault(id(im), im.to(dev).to(torch.bfloat16)) for im in host_imgs]
o = eng.generate(ids, images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=1, max_new_tokens=int(os.environ.get("NEW_TOKENS", "6")))
torch.cuda.synchronize()
print(o.stats.get("ctx_tokens_distinct"), o.stats.get("ctx_tokens_rows"), o.stats.get("n_groups"))
