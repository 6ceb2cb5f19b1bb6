import torch
p=torch.nn.Parameter(torch.randn(64,64,device="cuda"))
opt=torch.optim.AdamW([p],lr=1e-3,weight_decay=1e-4,amsgrad=True,fused=True)
for i in range(3):
    p.grad=torch.randn_like(p)
    v0=p._version; opt.step(); print("fused step", i, v0, "->", p._version)
