"""Micro-reproduction attempt (torch only, none of this package's kernels): do torch's multi-block reductions (max / sum / mean over
millions of elements: partials buffer + semaphores zeroed by a memsetAsync in front of the kernel) return stale or foreign values when
they are replayed from a HIP graph?  tools/diag_real_graph.py saw exactly that signature in the replayed discriminator update
(`decoded.abs().max()` frozen at the previous replay's value, `reals.abs().max()` returning max|decoded|).
Prints one JSON line per mode: how many of the replayed reduction outputs differ from the eager evaluation of the same input.
usage: python tools/diag_graph_reduce.py [out.jsonl]"""
import json
import sys
import time

import torch

OUT = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
T0 = time.time()


def emit(o):
    o["t"] = round(time.time() - T0, 1)
    with open(OUT, "a") as f:
        f.write(json.dumps(o) + "\n")


def body(inp, w, iters, with_backward):
    outs = []
    y = inp
    for k in range(iters):
        y2 = y * 1.0001 + 0.001 * k                  # a fresh temporary per iteration: allocator churn inside the capture
        outs.append(y2.abs().max())
        outs.append(torch.isnan(y2).sum())
        outs.append(torch.relu(1 - y2).mean())
        y = y2
    if with_backward:
        if w.grad is not None:
            w.grad.zero_()
        loss = torch.relu(1 - inp * w).mean() + torch.relu(1 + inp * w * 0.5).mean()
        loss.backward()
        outs.append(loss.detach())
        outs.append(w.grad.abs().max())
    return outs


def run(name, iters, replays, churn, with_backward, n=4 << 20):
    dev = torch.device("cuda", 0)
    g0 = torch.Generator(device=dev).manual_seed(1)
    src = [torch.randn(n, device=dev, generator=g0) * s for s in (0.1, 0.5, 0.25)]
    static = src[0].clone()
    w = torch.ones(n, device=dev, requires_grad=True)
    body(static, w, iters, with_backward)            # eager warm-up (lazy init, autograd thread)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = body(static, w, iters, with_backward)
    bad_replays, bad_values, first = 0, 0, []
    for r in range(replays):
        static.copy_(src[r % 3] * (1.0 + 0.01 * r))
        if churn:
            junk = [torch.randn((8 + (r + j) % 5) << 20, device=dev) for j in range(12)]
            s = sum(float(j.abs().max()) for j in junk[:2])      # eager reductions + host syncs between replays
            del junk, s
        graph.replay()
        got = [float(o) for o in outs]
        want = [float(o) for o in body(static, w, iters, with_backward)]
        diff = [i for i, (a, b) in enumerate(zip(got, want)) if not (a == b or abs(a - b) <= 1e-6 * max(abs(b), 1e-6))]
        if diff:
            bad_replays += 1
            bad_values += len(diff)
            if len(first) < 4:
                first.append({"replay": r, "n_bad": len(diff), "examples": [(i, got[i], want[i]) for i in diff[:4]]})
    emit({"mode": name, "iters": iters, "outputs": len(outs), "replays": replays, "bad_replays": bad_replays, "bad_values": bad_values, "first": first})


def main():
    emit({"torch": torch.__version__, "hip": torch.version.hip})
    for name, iters, replays, churn, bwd in (("plain", 40, 40, False, False), ("churn", 40, 40, True, False),
                                            ("churn+backward", 40, 40, True, True), ("long graph", 400, 12, True, True)):
        try:
            run(name, iters, replays, churn, bwd)
        except Exception as e:      # noqa: BLE001
            emit({"mode": name, "error": repr(e)[:400]})
    emit({"done": True})


if __name__ == "__main__":
    main()
