"""How well CAN the solo workers of the balanced kernel be loaded on bench.py's ragged batch (cfg3, seq_lens ~ U{1..1024},
seed 4321)?  No GPU: replays the kernel's static hand-out (pa_queue.hpp: first round in index order, the rest by the snake
over the ranks, keyed by each worker's place among its first-round peers) and prints the spread of the workers' token sums.
With 3072 items on 1536 workers every worker holds exactly two items, and pairing the longest first item with the shortest
second one — what the snake does — is the pairing that MINIMISES the largest pair sum; what is left is the spread of those
sums, a property of the batch, not of the schedule."""
import numpy as np
import torch

g = torch.Generator(device="cpu").manual_seed(4321)
lens = torch.randint(1, 1025, (256,), generator=g, dtype=torch.int32).numpy()
lens[0] = 1024
H, W = 12, 1536
R0 = W // H
first, rest = lens[:R0], lens[R0:]
order = np.argsort(-rest, kind="stable")
sums = []
for w in range(W):
    s0, h = divmod(w, H)
    ahead = int(np.sum((first > first[s0]) | ((first == first[s0]) & (np.arange(R0) < s0))))
    t = W - 1 - (ahead * H + h)
    sums.append(int(first[s0]) + int(rest[order[t // H]]))
sums = np.array(sums)
print(f"tokens per worker: mean {sums.mean():.0f}, min {sums.min()}, max {sums.max()}  ->  max / mean = {sums.max() / sums.mean():.3f}")
both = np.sort(lens)[::-1]
pair = both[: len(both) // 2] + both[::-1][: len(both) // 2]
print(f"all 256 ranked up front (no first round in index order): max / mean = {pair.max() / pair.mean():.3f}")
for fixed in (2.3, 3.4):
    cost = 2 * fixed + sums * 0.056          # per item: fixed + 56 ns/token (profiles/r02j_queue_cost_probe.log)
    print(f"cost model {fixed} us + 56 ns/token per item: slowest worker {cost.max():.1f} us, mean {cost.mean():.1f} us")
