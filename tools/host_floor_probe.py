"""Where does the host time of one GAN cycle go, and what does a hipGraph replay of it cost the host?

  python tools/host_floor_probe.py profile            cProfile of eager cycles (top functions by own time)
  python tools/host_floor_probe.py graph [1stream]    capture the cycle, time graph.replay() on the host and end to end
  python tools/host_floor_probe.py eager [1stream]    eager cycles: host enqueue ms and end-to-end ms
Each mode is one process (run it under the env flags to compare: DEBUG_HIP_FORCE_GRAPH_QUEUES, DEBUG_HIP_GRAPH_BATCH_SIZE,
DEBUG_CLR_GRAPH_PACKET_CAPTURE ...)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    one = "1stream" in sys.argv[2:]
    import torch
    import bench
    from hific_amd import ops
    if one:
        ops.set_side_stream(False)
        ops.set_branch_streams(False)
    args = argparse.Namespace(batch=16, size=256, dtype="bf16", regime="low", seed=0, steps=8, warmup=3)
    dev = torch.device("cuda:0")
    model, opts, reducers = bench.build(args, dev, "gan")
    step = bench.make_step(args, model, opts, reducers, dev, "gan")
    tag = f"{mode}{' 1stream' if one else ''} " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DEBUG_"))

    def fence():
        torch.cuda.synchronize()

    def measure(fn, n=8):
        fence(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        th = time.perf_counter() - t0
        fence(); te = time.perf_counter() - t0
        return th / n * 1e3, te / n * 1e3

    for _ in range(3):
        step()
    fence()
    if mode == "eager":
        h, e = measure(step)
        print(f"[probe] {tag}: host enqueue {h:.2f} ms/cycle, end-to-end {e:.2f} ms/cycle", flush=True)
    elif mode == "profile":
        import cProfile
        import pstats
        pr = cProfile.Profile()
        fence()
        pr.enable()
        for _ in range(4):
            step()
        pr.disable()
        fence()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(45)
        st.sort_stats("cumulative").print_stats(40)
    elif mode == "graph":
        from hific_amd.graph import GraphedStep
        t0 = time.perf_counter()
        gs = GraphedStep(step, warmup=2, generators=step.generators)
        print(f"[probe] {tag}: capture+instantiate {time.perf_counter() - t0:.2f} s", flush=True)
        gs(); fence()
        h, e = measure(gs)
        print(f"[probe] {tag}: hipGraphLaunch host {h:.2f} ms/replay, end-to-end {e:.2f} ms/cycle", flush=True)


if __name__ == "__main__":
    main()
