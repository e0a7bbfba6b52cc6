"""Bisects hipGraph capture problems: each case captures a progressively larger piece of the training step in its own
process (a runtime crash inside hipStreamEndCapture must not take the other cases down).
  python tools/graph_probe.py            -> runs every case in a subprocess, prints one line per case
  python tools/graph_probe.py --case N   -> runs case N in this process"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ["syn_fork_alloc", "syn_fork_twice", "syn_record_stream", "syn_event_later", "syn_fork_hific", "model_fwd_mask1",
         "model_fwd_mask2", "model_fwd_mask4", "model_fwd_mask3", "gturn_mask1", "gturn_mask2", "gturn_mask4", "model_fwd_mask5", "model_fwd_mask6", "model_fwd_mask7", "gturn_mask7",
         "add", "conv_fwd_small", "conv_fwd_sp9", "conv_fwd_bwd_1stream", "conv_fwd_bwd_side", "encoder_fwd",
         "model_fwd_1stream", "model_fwd_branch", "gturn_1stream", "gturn_side", "gturn_branch", "gturn_all",
         "cycle_1stream", "cycle_all"]


def run_case(name):
    import torch
    import hific_amd
    from hific_amd import ops, lib, optim
    from hific_amd.graph import GraphedStep
    from hific_amd.default_config import make_args, hific_args, ModelTypes
    dev = torch.device("cuda:0")
    hific_amd.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    side = name.endswith("_side") or name.endswith("_all")
    branch = name.endswith("_branch") or name.endswith("_all")
    ops.set_side_stream(side)
    ops.set_branch_streams(branch)
    if "_mask" in name:
        ops.set_branch_streams(True)
        ops.set_branch_mask(int(name.split("_mask")[1]))
        name = name.split("_mask")[0] + "_x"
    if name.startswith("syn_"):
        s2 = torch.cuda.Stream()
        a = torch.randn(1 << 20, device=dev)
        keep = {}

        def fn():
            main = torch.cuda.current_stream()
            s2.wait_stream(main)
            if name == "syn_record_stream":
                a.record_stream(s2)
            with torch.cuda.stream(s2):
                b = (a * 2.0 + 1.0) if name != "syn_fork_hific" else ops._add(a, a)
                if name == "syn_event_later":
                    keep["ev"] = torch.cuda.current_stream().record_event()
            c = a + 3.0
            if name == "syn_event_later":
                main.wait_event(keep.pop("ev"))
            else:
                main.wait_stream(s2)
            if name == "syn_record_stream":
                b.record_stream(main)
            out = b + c
            if name == "syn_fork_twice":
                s2.wait_stream(main)
                with torch.cuda.stream(s2):
                    d = out * 0.5
                main.wait_stream(s2)
                out = out + d
            return out
    elif name == "add":
        a = torch.randn(1 << 20, device=dev).bfloat16()
        fn = lambda: ops._add(a, a)
    elif name.startswith("conv_fwd_small") or name.startswith("conv_fwd_sp9") or name.startswith("conv_fwd_bwd"):
        C = 960 if "sp9" in name else 64
        x = torch.randn(4, C, 16, 16, device=dev).bfloat16().requires_grad_(True)
        w = torch.nn.Parameter(torch.randn(C, C, 3, 3, device=dev) * 0.02)
        b = torch.nn.Parameter(torch.zeros(C, device=dev))
        opt = optim.FusedAdam([w, b], lr=1e-4)

        def fn():
            y = ops.conv2d(x, w, b, 1, (1, 1, 1, 1), lib.PAD_REFLECT, act="relu")
            if "bwd" in name:
                y.float().square().mean().backward()
                opt.step(); opt.zero_grad()
            return y
    else:
        args = make_args(hific_args, n_residual_blocks=2, batch_size=4, image_dims=(3, 128, 128), latent_dims=(220, 8, 8))
        model = hific_amd.Model(args, model_type=ModelTypes.COMPRESSION_GAN, device_rate_select=True,
                                allow_random_lpips_backbone=True).to(dev).train()
        amort = [p for m in model.amortization_models for p in m.parameters()]
        opts = {"amort": optim.FusedAdam(amort, lr=1e-4),
                "hyper": optim.FusedAdam(list(model.Hyperprior.hyperlatent_likelihood.parameters()), lr=1e-4),
                "disc": optim.FusedAdam(list(model.Discriminator.parameters()), lr=1e-4)}
        x = torch.rand(4, 3, 128, 128, device=dev)
        if name == "encoder_fwd":
            def fn():
                with torch.no_grad():
                    return model.Encoder(x)
        elif name.startswith("model_fwd"):
            def fn():
                with torch.no_grad():
                    return model(x, train_generator=True, writeout=False)["compression"]
        elif name.startswith("gturn"):
            def fn():
                losses = model(x, train_generator=True, writeout=False)
                losses["compression"].backward()
                for n in ("amort", "hyper"):
                    opts[n].step(); opts[n].zero_grad()
                opts["disc"].zero_grad()
                return losses["compression"].detach()
        else:
            def fn():
                losses = model(x, train_generator=True, writeout=False)
                losses["compression"].backward()
                for n in ("amort", "hyper"):
                    opts[n].step(); opts[n].zero_grad()
                losses = model(x, train_generator=False, writeout=False)
                losses["disc"].backward()
                opts["disc"].step(); opts["disc"].zero_grad(); opts["amort"].zero_grad(); opts["hyper"].zero_grad()
                return losses["disc"].detach()
    gs = GraphedStep(fn, warmup=2)
    print(f"  {name}: captured", flush=True)
    for _ in range(3):
        out = gs()
    torch.cuda.synchronize()
    o = out if torch.is_tensor(out) else out[0]
    print(f"  {name}: replayed x3, finite={bool(torch.isfinite(o.float()).all())}", flush=True)


if __name__ == "__main__":
    if "--case" in sys.argv:
        run_case(sys.argv[sys.argv.index("--case") + 1])
    else:
        only = [a for a in sys.argv[1:] if a in CASES] or CASES
        for c in only:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--case", c], capture_output=True, text=True,
                               timeout=300)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip()]
            status = "OK" if r.returncode == 0 else f"FAIL rc={r.returncode}"
            msg = tail[-1][:200] if tail else ""
            err = next((l for l in tail if "Error" in l or "error" in l), "")[:200]
            print(f"{c:24s} {status:14s} {msg if r.returncode == 0 else err or msg}", flush=True)
