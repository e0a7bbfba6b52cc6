"""TEST INFRASTRUCTURE - recipe for `oracle/_ref/` (git-ignored, travels to the GPU box with gpurun).

The reference is Python: nothing to compile.  For the one GPU test that needs the reference's OWN stitcher code on
the device (`tests/test_gpu_injected_reference.py`: `inject.patch_reference()` -> the reference's `src.model.Model`
runs on the HIP modules), this packs the reference's Python sources - read where they lie under /root/reference,
never copied into the repository's history - into ONE archive, `oracle/_ref/reference_src.tar.gz`.  The test unpacks
it into a temporary directory at run time.  No weights, no data, no bytecode.

    python oracle/make_ref.py            (also run by __graft_entry__.build() when /root/reference exists)
"""
import os
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HIFIC_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref", "reference_src.tar.gz")


def main():
    if not os.path.isdir(os.path.join(REF, "src")):
        print(f"make_ref: {REF}/src not present - nothing to do")
        return 1
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    n = 0
    with tarfile.open(OUT, "w:gz") as tar:
        tar.add(os.path.join(REF, "default_config.py"), arcname="default_config.py")
        for root, dirs, files in os.walk(os.path.join(REF, "src")):
            dirs[:] = [d for d in dirs if d != "__pycache__"]
            for f in sorted(files):
                if f.endswith(".py"):
                    full = os.path.join(root, f)
                    tar.add(full, arcname=os.path.relpath(full, REF))
                    n += 1
    print(f"make_ref: {n + 1} source files -> {OUT} ({os.path.getsize(OUT)} bytes)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
