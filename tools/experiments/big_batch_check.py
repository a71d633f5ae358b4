import sys, torch
sys.path.insert(0, "/root/repo")
from calipsync_amd import recipe
from calipsync_amd.unet import Model
net = Model(6, "hubert").to("cuda:0")
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
x16, a16 = recipe.make_inputs(16)
for B in (1024, 2048):
    x = torch.from_numpy(x16).cuda().repeat(B // 16, 1, 1, 1); a = torch.from_numpy(a16).cuda().repeat(B // 16, 1, 1, 1)
    out = net(x, a); torch.cuda.synchronize()
    ref = net(x[:16].contiguous(), a[:16].contiguous())
    d = max(float((out[i*16:(i+1)*16] - ref).abs().max()) for i in (0, B//32, B//16 - 1))
    print(B, "max diff vs 16-frame forward:", d, "finite:", bool(torch.isfinite(out).all()), "mem GB", torch.cuda.max_memory_allocated()/1e9)
    del out, x, a; torch.cuda.empty_cache()
