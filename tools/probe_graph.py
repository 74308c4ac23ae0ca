"""Why does hipGraph replay of the training step lose to eager launches (bench.py: probe_graph_ms 33.6 vs probe_eager_ms 19.2 in round 4)?

Captures the Poseidon-B batch-64 step as bench.py does, in three forms — with the weight-gradient stream forked / joined through events
(the default), with everything on ONE stream (SCOT_SIDE_STREAM=0), and the forward alone — dumps the captured graph
(hipGraphDebugDotPrint through torch's debug_dump), counts nodes / edges / fan-in, and times replay against eager launches of the same
program.  One process per form (the engine reads its switches at construction).

    python tools/probe_graph.py [--out gpurun_out/graph] [--form side|single|fwd]     (no --form: runs all three as subprocesses)
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dot_stats(path):
    txt = open(path, errors="replace").read()
    nodes = re.findall(r'^\s*"?([\w.]+)"?\s*\[(.*?)\];', txt, flags=re.M | re.S)
    edges = re.findall(r'^\s*"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt, flags=re.M)
    kinds = {}
    for _, attr in nodes:
        m = re.search(r'label\s*=\s*"([^"]*)"', attr)
        lab = (m.group(1) if m else "?")
        kind = "kernel" if ("KERNEL" in lab.upper() or "(" in lab) else lab.split("\\n")[0].split(" ")[0][:24]
        for key in ("MEMSET", "MEMCPY", "EVENT_RECORD", "WAIT_EVENT", "EMPTY", "HOST"):
            if key in lab.upper():
                kind = key.lower()
        kinds[kind] = kinds.get(kind, 0) + 1
    fan_in, fan_out = {}, {}
    for a, b in edges:
        fan_out[a] = fan_out.get(a, 0) + 1
        fan_in[b] = fan_in.get(b, 0) + 1
    return {"nodes": len(nodes), "edges": len(edges), "kinds": dict(sorted(kinds.items(), key=lambda kv: -kv[1])[:8]),
            "nodes_with_fan_in_ge_2": sum(1 for v in fan_in.values() if v >= 2), "nodes_with_fan_out_ge_2": sum(1 for v in fan_out.values() if v >= 2),
            "dot_bytes": len(txt)}


def run_form(form, out):
    import torch
    from poseidon_amd.config import preset
    from scOT.model import ScOT
    torch.manual_seed(0)
    cfg = preset("B", image_size=128, num_channels=4, num_out_channels=4, channel_slice_list_normalized_loss=[0, 1, 3, 4])
    model = ScOT(cfg, compute="fp16").to("cuda")
    B = 64
    kw = dict(pixel_values=torch.randn(B, 4, 128, 128, device="cuda"), time=torch.rand(B, device="cuda"),
              labels=torch.randn(B, 4, 128, 128, device="cuda"))

    def step():
        if form == "fwd":
            with torch.no_grad():
                model(**kw)
            return
        model.zero_grad(overlap=True)
        model(**kw).loss.backward()

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        enq = (time.perf_counter() - t) / n
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3, enq * 1e3

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    eager_ms, eager_enq = timed(step)
    # what a captured graph of this program contains: launches and cross-stream edges, counted from the engine's own recorded step
    prog = {}
    for ent in model._engine._taped.values():
        for part in ("fwd", "bwd"):
            for fn, _ in ent.get(part) or []:
                name = getattr(fn, "__name__", None) or "host"
                kind = "event_record" if name == "scot_event_record" else "stream_wait_event" if name == "scot_stream_wait_event" else \
                    "memset/memcpy" if name in ("scot_memset_async", "scot_memcpy_async") else "host" if name in ("host", "run", "<lambda>") else "launch"
                prog[kind] = prog.get(kind, 0) + 1
    g = torch.cuda.CUDAGraph()
    g.enable_debug_mode()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    dot = os.path.abspath(os.path.join(out, f"step_{form}.dot"))
    stats = None
    try:
        g.debug_dump(dot)
        stats = dot_stats(dot)
        if stats["dot_bytes"] > 3 << 20:      # keep the numbers, not megabytes of node labels
            os.remove(dot)
    except Exception as e:  # pragma: no cover
        stats = {"error": repr(e)}
    graph_ms, graph_enq = timed(g.replay)
    res = {"form": form, "recorded_program": prog, "eager_ms": eager_ms, "eager_enqueue_ms": eager_enq, "graph_ms": graph_ms, "graph_enqueue_ms": graph_enq, "graph": stats,
           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/graph")
    ap.add_argument("--form", default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.form:
        return run_form(a.form, a.out)
    rows = []
    for form, env in (("side", {}), ("single", {"SCOT_SIDE_STREAM": "0"}), ("fwd", {}), ("side", {"GPU_MAX_HW_QUEUES": "2"}),
                      ("side", {"DEBUG_HIP_GRAPH_DOT_PRINT": "0", "HIP_GRAPH_BRANCH_STREAMS": "1"})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--out", a.out, "--form", form], env={**os.environ, **env},
                           capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        row = json.loads(line[-1]) if line else {"form": form, "error": p.stderr[-800:]}
        row["env"] = env
        rows.append(row)
        print(json.dumps(row), flush=True)
    json.dump(rows, open(os.path.join(a.out, "graph_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
