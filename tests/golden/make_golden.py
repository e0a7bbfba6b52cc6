"""Generates tests/golden/*.pt from the REAL reference (imported from /root/reference via oracle/ref_loader.py).
Run in the build container only:   python tests/golden/make_golden.py
Every fixture stores the seeds that regenerate its inputs/weights (oracle.make_state_dict / make_image / make_noise)
and small outputs of the reference itself; tests/test_oracle_golden.py replays them anywhere (no reference needed).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader, hific_oracle as O  # noqa: E402


def grad_norms(named_params, keys):
    d = dict(named_params)
    return {k: float(d[k].grad.norm()) for k in keys}


WATCH = ["Encoder.conv_block1.1.weight", "Encoder.conv_block4.2.gamma", "Generator.resblock_3.conv1.weight",
         "Generator.upconv_block2.0.weight", "Generator.conv_block_out.1.bias",
         "Hyperprior.analysis_net.conv2.weight", "Hyperprior.synthesis_mu.conv1.weight",
         "Hyperprior.synthesis_std.conv3.weight", "Hyperprior.hyperlatent_likelihood.H_1",
         "Hyperprior.hyperlatent_likelihood.a_2", "Hyperprior.hyperlatent_likelihood.b_0"]


def model_golden(ns, gan, training, train_generator=True):
    torch.manual_seed(0)
    m = ref_loader.build_reference_model(ns, gan=gan, training=training)
    sd = O.make_state_dict(seed=0, gan=gan)
    m.load_state_dict(sd, strict=True)
    ref_loader.set_lpips_backbone(m, O.make_alex_backbone())
    x = O.make_image(1, 2, 128, 128)
    nh, nl = O.make_noise(6, (2, 320, 2, 2)), O.make_noise(7, (2, 220, 8, 8))
    noises = [nh, nl]
    orig = ns.hyperprior.CodingModel._quantize

    def patched(self, x_, mode='noise', means=None):
        if mode == 'noise':
            return x_ + noises.pop(0)
        return orig(self, x_, mode=mode, means=means)

    ns.hyperprior.CodingModel._quantize = patched
    try:
        m.step_counter = 0
        losses, inter = m(x, train_generator=train_generator, return_intermediates=True, writeout=False)
    finally:
        ns.hyperprior.CodingModel._quantize = orig
    out = dict(gan=gan, training=training, train_generator=train_generator,
               seeds=dict(sd=0, image=1, noise_h=6, noise_l=7, B=2, H=128),
               compression=float(losses["compression"]), n_bpp=float(inter.n_bpp), q_bpp=float(inter.q_bpp),
               recon_patch=inter.reconstruction[:, :, :6, :6].detach().clone(),
               recon_mean=float(inter.reconstruction.mean()), recon_std=float(inter.reconstruction.std()),
               latents_sum=float(inter.latents_quantized.sum()),
               latents_patch=inter.latents_quantized[:, :4, :3, :3].detach().clone())
    if training:
        key = "compression" if train_generator else "disc"
        losses[key].backward()
        watch = list(WATCH) if train_generator else []
        if gan:
            out["disc"] = float(losses["disc"])
            watch += ["Discriminator.conv2.weight_orig", "Discriminator.context_conv.weight"]
            out["weight_u_after"] = m.Discriminator.conv3.weight_u.detach().clone()
        out["grad_norms"] = grad_norms(m.named_parameters(), watch)
    return out


def main():
    ns = ref_loader.load()
    g = {}
    g["compression_train"] = model_golden(ns, gan=False, training=True)
    g["compression_eval"] = model_golden(ns, gan=False, training=False)
    g["gan_train_G"] = model_golden(ns, gan=True, training=True, train_generator=True)
    g["gan_train_D"] = model_golden(ns, gan=True, training=True, train_generator=False)
    # primitive-level vectors from the reference modules
    x = O.make_noise(11, (2, 12, 5, 7)) * 3
    cn = ns.channel.ChannelNorm2D(12)
    with torch.no_grad():
        cn.gamma.copy_(O.make_noise(12, (1, 12, 1, 1)) + 1.2)
        cn.beta.copy_(O.make_noise(13, (1, 12, 1, 1)))
    g["channelnorm"] = dict(x_seed=11, y=cn(x).detach().clone())
    lb = ns.maths.LowerBoundToward.apply
    t = (O.make_noise(14, (64,)) * 2).requires_grad_(True)
    y = lb(t, 0.11)
    gg = O.make_noise(15, (64,))
    y.backward(gg)
    g["lower_bound"] = dict(y=y.detach().clone(), dx=t.grad.clone())
    g["sched"] = [float(ns.utils.get_scheduled_params(2.0, dict(vals=[2., 1.], steps=[50000]), s)) for s in
                  (0, 1, 49999, 50000, 70000)]
    torch.save(g, os.path.join(HERE, "reference_outputs.pt"))
    print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if isinstance(vv, float)})
           for k, v in g.items()})


if __name__ == "__main__":
    main()
