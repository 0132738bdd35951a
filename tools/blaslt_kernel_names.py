import torch
dev="cuda:0"
for (M,N,K) in ((39168,12288,4096),(8192,8192,8192),(1536,12288,4096),(1536,4096,11008)):
    x=torch.randn(M,K,device=dev).bfloat16(); w=torch.randn(N,K,device=dev).bfloat16()
    for _ in range(3): y=torch.matmul(x,w.t())
torch.cuda.synchronize()
