import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
from ref_llava import RefLlava
from test_engine_gpu import prompts
DEV='cuda:0'
cfg=preset('tiny'); w=LlavaWeights.random(cfg,DEV,seed=3,std=0.06)
e=VddLlavaEngine(cfg,weights=w,device=DEV,t_max=256,use_graph=False)
ref=RefLlava(w,device=DEV, logit_dtype=torch.float32)
ids,imgs=prompts()
for share in (True,False):
    out=e.generate(ids,images=imgs,cd_alpha=1.0,cd_beta=0.1,temperature=0.5,max_new_tokens=2,cd_greedy=True,use_dd=True,use_dd_unk=True,share_prefix=share)
    L=e.debug_logits0.float().cpu(); Q=len(ids)
    for q in range(Q):
        i=ids[q]
        m=ref(input_ids=i[None], images=imgs[q][None]).logits[0,-1]
        u=i.clone(); u[u==-200]=0
        un=ref(input_ids=u[None], images=None).logits[0,-1]
        n=i[i!=-200]
        no=ref(input_ids=n[None], images=None).logits[0,-1]
        print(share,q,'main',(L[q]-m).abs().max().item(),'unk',(L[Q+q]-un).abs().max().item(),'none',(L[2*Q+q]-no).abs().max().item(),'scale',m.abs().max().item())
