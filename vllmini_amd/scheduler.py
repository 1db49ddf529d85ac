"""Continuous-batching decode scheduler over the paged-attention ops.

Counterpart of the reference's `Scheduler` (vllmini/scheduler.py:10-156).  Same life cycle and
same per-sequence semantics:

    add_sequence(input_ids)   prefill on arrival, keep the last-position logits   (scheduler.py:22-53)
    run() / step()            sample the next token from the stored logits (temperature 1, top-k 50,
                              multinomial; :144-153), append it, run one decode step, store the new
                              logits, stop at EOS or max_length and free the blocks (:76-108)
    out of KV blocks          evict the YOUNGEST other sequence and carry on (:117-130)

What differs is the batch dimension the reference never uses: its loop takes ONE sequence id from a
priority queue per iteration (:60) although paged_attention_v1 / reshape_and_cache are batched.  Here
every step advances all running sequences (oldest first, up to max_batch) in one call of
GPT2PagedDecoder.decode — one reshape_and_cache + one paged_attention_v1 per layer for the whole batch.

Across GPUs (vllmini_amd/shard.py): one scheduler per rank over a private KV pool; requests are dealt to
ranks by arrival index, and the only per-step exchange is the all_gather of the sampled token ids.

The scheduler only needs an object with prefill(seq_id, ids) -> logits[V] and
decode(seq_ids, tokens) -> logits[B, V] plus a `.pool` with free()/RuntimeError on exhaustion, so its
logic is testable on CPU with a stand-in decoder (tests/test_scheduler.py); on the GPU it drives
GPT2PagedDecoder (tests/test_parity_gpu.py).
"""
from __future__ import annotations

import itertools
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


def sample_top_k(logits: torch.Tensor, top_k: int = 50, temperature: float = 1.0,
                 generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Batched form of Scheduler.sample_next_token (scheduler.py:144-153): [B, V] -> [B] int64.  Half logits on a HIP device
    (what the decoder hands over) are drawn by ONE launch of this build's kernel — the same distribution, the uniform numbers
    from the same torch generator (vllmini_amd/gpt2_layer.py sample_top_k: 127 / 309 us -> 14 for 1 / 256 rows); anything else
    takes the torch chain the reference spells out."""
    if logits.is_cuda and logits.dtype == torch.float16 and logits.dim() == 2 and logits.stride(1) == 1 \
            and logits.shape[1] <= 65536 and top_k <= 64 and (generator is None or generator.device.type == "cuda"):
        from . import gpt2_layer
        return gpt2_layer.sample_top_k(logits, top_k, temperature, generator)
    logits = logits.float() / temperature
    k = min(top_k, logits.shape[-1])
    vals, idx = torch.topk(logits, k, dim=-1)
    probs = F.softmax(vals, dim=-1)
    choice = torch.multinomial(probs, num_samples=1, generator=generator)
    return idx.gather(-1, choice).squeeze(-1)


def sample_greedy(logits: torch.Tensor, **_) -> torch.Tensor:
    return logits.argmax(-1)


class BatchScheduler:
    def __init__(self, decoder, max_length: int, eos_token_id: int, max_batch: int = 256,
                 sampler: Callable[..., torch.Tensor] = sample_top_k, use_graph: bool = False,
                 generator: Optional[torch.Generator] = None):
        self.decoder = decoder
        # a sequence may grow to max_length tokens; the pool's table must hold them (the reference crashes at that
        # point, block_manager.py:41-63 — here ONE such sequence would abort the step of every sequence in its batch)
        pool = getattr(decoder, "pool", None)
        if pool is not None and max_length > (pool.max_blocks_per_seq - 1) * pool.block_size:
            # (the table keeps a trailing -1, as the reference's last-block search needs it: block_manager.py:36-39)
            raise ValueError(f"max_length={max_length} does not fit (max_blocks_per_seq - 1) x block_size = "
                             f"{(pool.max_blocks_per_seq - 1) * pool.block_size} tokens")
        self.max_length = max_length                         # scheduler.py:15
        self.eos_token_id = eos_token_id
        self.max_batch = max_batch
        self.sampler = sampler
        self.use_graph = use_graph
        self.generator = generator
        self._ids = itertools.count()
        self.active: Dict[int, int] = {}                     # seq_id -> arrival index   (active_sequences, :17)
        self.last_logits: Dict[int, torch.Tensor] = {}       # :18
        self.sequence_lengths: Dict[int, int] = {}           # :19
        self.sequences: Dict[int, List[int]] = {}            # :20 (kept after completion for result polling)
        self.evicted: List[int] = []
        self.steps = 0

    # ---- arrival: prefill immediately (scheduler.py:22-53) --------------------------------------------
    def add_sequence(self, input_ids: Sequence[int]) -> int:
        seq_id = next(self._ids)
        ids = [int(t) for t in input_ids]
        while True:
            try:
                logits = self.decoder.prefill(seq_id, ids)
                break
            except RuntimeError as e:
                if "free blocks" not in str(e) or not self._evict_youngest(exclude=()):
                    raise
        self.active[seq_id] = seq_id
        self.last_logits[seq_id] = logits
        self.sequence_lengths[seq_id] = len(ids)
        self.sequences[seq_id] = ids
        return seq_id

    # ---- one decode step for every running sequence ------------------------------------------------------
    def _finish(self, seq_id: int) -> None:                  # remove_sequence_from_processing, :132-138
        self.decoder.pool.free(seq_id)
        self.active.pop(seq_id, None)
        self.last_logits.pop(seq_id, None)
        self.sequence_lengths.pop(seq_id, None)

    def _evict_youngest(self, exclude: Sequence[int]) -> bool:
        """handle_out_of_memory (:117-130): drop the most recently arrived sequence not in `exclude`,
        falling back to the youngest overall."""
        if not self.active:
            return False
        cands = [s for s in self.active if s not in exclude] or list(self.active)
        victim = max(cands, key=self.active.get)
        self._finish(victim)
        self.evicted.append(victim)
        return True

    def step(self) -> List[int]:
        """Advance up to max_batch running sequences by one token; returns the ids that were stepped."""
        # sequences already at max_length end without sampling (:71-74)
        for sid in [s for s in self.active if self.sequence_lengths[s] >= self.max_length]:
            self._finish(sid)
        batch = sorted(self.active, key=self.active.get)[: self.max_batch]      # oldest first (PriorityQueue, :16)
        if not batch:
            return []
        logits = torch.stack([self.last_logits[s] for s in batch])
        tokens = self.sampler(logits, generator=self.generator)                 # :76
        tok_list = tokens.tolist()
        while True:
            try:
                new_logits = self.decoder.decode(batch, tokens, use_graph=self.use_graph) \
                    if self.use_graph else self.decoder.decode(batch, tokens)
                break
            except RuntimeError as e:                                           # :110-115
                if "free blocks" not in str(e):
                    raise
                # the reference evicts the youngest sequence other than the one being processed; with a
                # batch in flight, prefer a victim outside the batch, else the youngest member
                outside = [s for s in self.active if s not in batch]
                victim = max(outside or batch, key=self.active.get)
                self._finish(victim)
                self.evicted.append(victim)
                if victim in batch:
                    keep = [i for i, s in enumerate(batch) if s != victim]
                    batch = [batch[i] for i in keep]
                    tokens = tokens[keep]
                    tok_list = [tok_list[i] for i in keep]
                if not batch:
                    return []
                # NB: a step that failed mid-way may have advanced some rows' bookkeeping; the pool hands
                # out blocks before touching fill counters, so retrying is consistent (kv_pool._step_rows)
        for i, sid in enumerate(batch):
            self.sequences[sid].append(tok_list[i])                             # :79
            self.last_logits[sid] = new_logits[i]                               # :100
            self.sequence_lengths[sid] += 1                                     # :101
            if tok_list[i] == self.eos_token_id or self.sequence_lengths[sid] >= self.max_length:   # :103-108
                self._finish(sid)
        self.steps += 1
        return batch

    def run(self, max_steps: Optional[int] = None) -> int:
        """Step until no sequence is running (scheduler.py:55-115); returns the number of steps."""
        n = 0
        while self.active and (max_steps is None or n < max_steps):
            self.step()
            n += 1
        return n


def deal_requests(num_requests: int, rank: int, world: int) -> List[int]:
    """Arrival index -> rank, round robin (keeps each rank's batch the same age mix)."""
    return [i for i in range(num_requests) if i % world == rank]
