#!/usr/bin/env python3
"""bench.py under `rocprofv3 --pmc ...`, with the process' memory map saved first.

Round 1 saw one host-side SIGSEGV inside a kernel launch when bench.py ran under the profiler's counter mode (frames in
libraries without symbols).  This driver writes /proc/self/maps next to the profile before the run, so that, should it
happen again, every frame of the profiler's own stack trace can be attributed to a library and an offset
(addr2line -e <lib> <frame - base>).  Usage (put python3 itself after `--`, no shell wrapper):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p -- python3 tools/pmc_bench.py <out_dir> [bench.py flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
    import torch  # noqa: F401  (libamdhip64, the HSA runtime and the profiler's tool library are mapped from here on)
    from raw_ngp_amd import _lib
    _lib.load()
    with open("/proc/self/maps") as f, open(os.path.join(out, "maps.txt"), "w") as g:
        g.writelines(l for l in f if " r-xp " in l or " r--p " in l)
    import bench
    bench.main()


if __name__ == "__main__":
    main()
