"""Per-(entry point, shape, stream) table of the launches inside replayed steps: python tools/launch_summary.py launches.json [top]
(the file `bench.py --launch-dump` writes: HIP-event pairs around every C-ABI call of the step tape, on the launching stream)."""
import json
import sys
from collections import defaultdict


def main():
    d = json.load(open(sys.argv[1]))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    n = d["nsteps"]
    acc = defaultdict(lambda: [0.0, 0])
    tot = [0.0, 0.0]
    for name, ints, is_main, t0, ms in d["rows"]:
        key = (name.replace("scot_", ""), tuple(x for x in ints if x > 1)[:8], is_main)
        acc[key][0] += ms
        acc[key][1] += 1
        tot[is_main] += ms
    print(f"steps {n}: main stream {tot[1] / n:.3f} ms/step of launches, other streams {tot[0] / n:.3f} ms/step")
    print(f"{'entry point':28s} {'stream':6s} {'calls/step':>10s} {'us each':>9s} {'ms/step':>8s}  shape ints")
    for (name, ints, is_main), (ms, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{name:28s} {'main' if is_main else 'side':6s} {c / n:10.1f} {ms / c * 1e3:9.1f} {ms / n:8.3f}  {list(ints)}")


if __name__ == "__main__":
    main()
