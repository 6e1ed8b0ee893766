import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synt_isic_amd.weights import synthetic_unet_state_dict
from synt_isic_amd.unet import HipUNet2DModel
from synt_isic_amd.scheduler import HipDDPMScheduler
from synt_isic_amd.train import HipAdam, HipGradScaler, mse_loss
from oracle import train as otrain
sd = synthetic_unet_state_dict()
g = torch.Generator().manual_seed(77)
images = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1); noise = torch.randn(2, 3, 64, 64, generator=g); ts = torch.tensor([37, 912])
_, ref_grads, _ = otrain.loss_and_grads(sd, images, noise, ts)
m = HipUNet2DModel(); m.load_state_dict(sd); m = m.to("cuda")
sch = HipDDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2")
opt = HipAdam(m.parameters(), lr=1e-4); sc = HipGradScaler(); m.train()
loss = mse_loss(m(sch.add_noise(images.cuda(), noise.cuda(), ts.cuda()), ts.cuda()).sample, noise.cuda())
opt.zero_grad(); sc.scale(loss).backward(); scaled = m.grads(); print("step ok", sc.step(opt)); sc.update()
after = m.state_dict(); ref_new, _ = otrain.adam_step(sd, ref_grads)
worst = None
for name in ref_new:
    mr = ref_new[name] - sd[name]; mv = after[name].cpu() - sd[name]
    clear = ref_grads[name].abs() > 1e-3
    if clear.any():
        d = (mv - mr).abs() * clear
        i = int(d.flatten().argmax()); v = float(d.flatten()[i])
        if worst is None or v > worst[0]:
            worst = (v, name, i, float(mv.flatten()[i]), float(mr.flatten()[i]), float(ref_grads[name].flatten()[i]), float(scaled[name].flatten()[i]) / 65536, float(sd[name].flatten()[i]))
print("worst (diff, name, idx, move_gpu, move_ref, g_ref, g_gpu, w):", worst)
nm = worst[1]
print("moves gpu", (after[nm].cpu()-sd[nm]).flatten()[:8], "ref", (ref_new[nm]-sd[nm]).flatten()[:8])
print("grads gpu", (scaled[nm]/65536).flatten()[:8], "ref", ref_grads[nm].flatten()[:8])
st = m.optimizer_state(); print("step", st["step"], "m", st["exp_avg"][nm].flatten()[:4], "v", st["exp_avg_sq"][nm].flatten()[:4])
