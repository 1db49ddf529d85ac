"""Continuous-batching decode scheduler over the paged-attention ops.

Counterpart of the reference's `Scheduler` (vllmini/scheduler.py:10-156).  Same life cycle and
same per-sequence semantics:

    add_sequence(input_ids)   prefill on arrival, keep the last-position logits   (scheduler.py:22-53)
    run() / step()            sample the next token from the stored logits (temperature 1, top-k 50,
                              multinomial; :144-153), append it, run one decode step, store the new
                              logits, stop at EOS or max_length and free the blocks (:76-108)
    out of KV blocks          handle_out_of_memory (:117-130) picks the YOUNGEST other sequence.  The reference drops it
                              (remove_sequence_from_processing: its tokens are lost) although its block manager can
                              move a sequence's pages to the host and back (block_manager.py:70-87, never called).
                              Here the victim is PREEMPTED: PagedKVPool.swap_out moves its pages to a pinned host pool in
                              one launch, the sequence keeps its logits and tokens, and it is swapped back in — oldest
                              first — as soon as the pool has room again, continuing bit-identically.  preempt="drop" is
                              the reference's behaviour (and the fallback when the host pool is full).

What differs is the batch dimension the reference never uses: its loop takes ONE sequence id from a
priority queue per iteration (:60) although paged_attention_v1 / reshape_and_cache are batched.  Here
every step advances all running sequences (oldest first, up to max_batch) in one call of
GPT2PagedDecoder.decode — one reshape_and_cache + one paged_attention_v1 per layer for the whole batch.

A serving loop on top of that (round 6; bench.py --serve measures it): submit() queues a request instead of prefilling it
at once, step() admits queued requests while a batch slot and blocks are free — several prompts through ONE prefill call
(decoder.prefill_batch) — and a request may carry its own max_new_tokens.  The host never waits for the GPU inside a step:
the sampled ids come back through a pinned buffer requested BEFORE the decode launches are queued, so the bookkeeping of
step k+1 runs while the GPU executes step k.

Across GPUs (vllmini_amd/shard.py): one scheduler per rank over a private KV pool; requests are dealt to
ranks by arrival index, and the only per-step exchange is the all_gather of the sampled token ids.

The scheduler only needs an object with prefill(seq_id, ids) -> logits[V] and
decode(seq_ids, tokens) -> logits[B, V] plus a `.pool` with free()/RuntimeError on exhaustion, so its
logic is testable on CPU with a stand-in decoder (tests/test_scheduler.py); on the GPU it drives
GPT2PagedDecoder (tests/test_parity_gpu.py, tests/test_serve_gpu.py).
"""
from __future__ import annotations

import collections
import itertools
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def sample_top_k(logits: torch.Tensor, top_k: int = 50, temperature: float = 1.0,
                 generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Batched form of Scheduler.sample_next_token (scheduler.py:144-153): [B, V] -> [B] int64.  Half logits on a HIP device
    (what the decoder hands over) are drawn by ONE launch of this build's kernel — the same distribution, the uniform numbers
    from the same torch generator (vllmini_amd/gpt2_layer.py sample_top_k: 127 / 309 us -> 14 for 1 / 256 rows); anything else
    — and a tree without libvmi_gpt2_layer.so — takes the torch chain the reference spells out.  The two draw from the same
    distribution with DIFFERENT random streams (inverse CDF over torch.rand against torch.multinomial): a seeded run
    reproduces itself, not the other path's tokens."""
    if logits.is_cuda and logits.dtype == torch.float16 and logits.dim() == 2 and logits.stride(1) == 1 \
            and logits.shape[1] <= 65536 and top_k <= 64 and (generator is None or generator.device.type == "cuda"):
        from . import gpt2_layer
        try:
            return gpt2_layer.sample_top_k(logits, top_k, temperature, generator)
        except gpt2_layer.LayerLibraryError:
            pass
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
                 generator: Optional[torch.Generator] = None, preempt: str = "swap", record_latency: bool = False,
                 max_prefill_tokens: int = 8192, admit_every: int = 1, headroom_blocks: Optional[int] = None):
        self.decoder = decoder
        # a sequence may grow to max_length tokens; the pool's table must hold them (the reference crashes at that
        # point, block_manager.py:41-63 — here ONE such sequence would abort the step of every sequence in its batch)
        pool = getattr(decoder, "pool", None)
        if pool is not None and max_length > (pool.max_blocks_per_seq - 1) * pool.block_size:
            # (the table keeps a trailing -1, as the reference's last-block search needs it: block_manager.py:36-39)
            raise ValueError(f"max_length={max_length} does not fit (max_blocks_per_seq - 1) x block_size = "
                             f"{(pool.max_blocks_per_seq - 1) * pool.block_size} tokens")
        if preempt not in ("swap", "drop"):
            raise ValueError("preempt must be 'swap' or 'drop'")
        self.max_length = max_length                         # scheduler.py:15
        self.eos_token_id = eos_token_id
        self.max_batch = max_batch
        self.sampler = sampler
        self.use_graph = use_graph
        self.generator = generator
        self.preempt = preempt if hasattr(pool, "swap_out") else "drop"
        self.max_prefill_tokens = max_prefill_tokens
        # queued requests are admitted every `admit_every` decode steps (or when nothing runs): 1 = at once, as the reference
        # prefills on arrival; larger = fewer, larger prefill calls (a prefill call is ~300 launches whatever it carries,
        # and at 2 sequences ending per step the slots and blocks of 16 steps make one call of ~32 prompts)
        self.admit_every = max(1, int(admit_every))
        self._last_admit_step = -(1 << 30)
        # blocks kept free per running sequence when a newcomer is let in (None: one per layer — a sequence's next block
        # boundary; 0: a prompt is admitted whenever it fits and growth is paid for by preemption, the reference's way)
        self.headroom_blocks = headroom_blocks
        self._ids = itertools.count()
        self.active: Dict[int, int] = {}                     # seq_id -> arrival index   (active_sequences, :17)
        self.last_logits: Dict[int, torch.Tensor] = {}       # :18 — of the sequences that were NOT in the last decode call
        self.sequence_lengths: Dict[int, int] = {}           # :19
        self.sequences: Dict[int, List[int]] = {}            # :20 (kept after completion for result polling)
        self.stop_at: Dict[int, int] = {}                    # per request: min(max_length, prompt + max_new_tokens)
        self.swapped: Dict[int, int] = {}                    # preempted, pages on the host: seq_id -> arrival index
        self.waiting: collections.deque = collections.deque()   # submitted, not prefilled yet: (seq_id, ids)
        self.evicted: List[int] = []                         # dropped for good (the reference's handle_out_of_memory)
        self.rejected: Dict[int, str] = {}                   # submitted requests that can never fit the pool: seq_id -> why
        self.steps = 0
        # the last decode call's logits stay ONE tensor: a step whose batch is the previous one samples from it directly
        self._lb_ids: List[int] = []
        self._lb_pos: Dict[int, int] = {}
        self._lb_logits: Optional[torch.Tensor] = None
        self.stats = {"preemptions": 0, "resumes": 0, "dropped": 0, "admitted": 0, "prefill_calls": 0, "decode_rows": 0,
                      "host_s": 0.0, "wait_s": 0.0, "admit_s": 0.0}
        self.record_latency = record_latency
        self._t_last: Dict[int, float] = {}
        self.token_latency_s: List[np.ndarray] = []          # per step: seconds since each stepped sequence's previous token
        self.first_token_s: List[float] = []                 # per request: submit / add -> first decode step done
        self._t_submit: Dict[int, float] = {}

    # ---- arrival -----------------------------------------------------------------------------------------
    def _new_request(self, input_ids: Sequence[int], max_new_tokens: Optional[int]) -> tuple:
        seq_id = next(self._ids)
        ids = [int(t) for t in input_ids]
        stop = self.max_length if max_new_tokens is None else min(self.max_length, len(ids) + int(max_new_tokens))
        self.stop_at[seq_id] = stop
        self.sequences[seq_id] = ids
        if self.record_latency:
            self._t_submit[seq_id] = time.perf_counter()
        return seq_id, ids

    def _started(self, seq_id: int, ids: List[int], logits: torch.Tensor) -> None:
        self.active[seq_id] = seq_id
        self.last_logits[seq_id] = logits
        self.sequence_lengths[seq_id] = len(ids)
        self.stats["admitted"] += 1

    def add_sequence(self, input_ids: Sequence[int], max_new_tokens: Optional[int] = None) -> int:
        """Prefill immediately (scheduler.py:22-53); out of blocks -> the youngest running sequence makes room."""
        seq_id, ids = self._new_request(input_ids, max_new_tokens)
        while True:
            try:
                logits = self.decoder.prefill(seq_id, ids)
                self.stats["prefill_calls"] += 1
                break
            except RuntimeError as e:
                if "free blocks" not in str(e) or not self._make_room(exclude=()):
                    self.sequences.pop(seq_id, None)
                    self.stop_at.pop(seq_id, None)
                    raise
        self._started(seq_id, ids, logits)
        return seq_id

    def submit(self, input_ids: Sequence[int], max_new_tokens: Optional[int] = None) -> int:
        """Queue a request; step() prefills it — with others, in one call — when a batch slot and its blocks are free."""
        seq_id, ids = self._new_request(input_ids, max_new_tokens)
        self.waiting.append((seq_id, ids))
        return seq_id

    def abort(self, seq_id: int) -> bool:
        """Cancel a request wherever it is — queued, running, or swapped out: its blocks (device or host) are free at once, its
        tokens so far stay readable in `sequences`.  False if the id is not pending.  (The reference has no cancellation; a
        serving front end needs one when a client goes away.)"""
        for i, (sid, _) in enumerate(self.waiting):
            if sid == seq_id:
                del self.waiting[i]
                self.stop_at.pop(seq_id, None)
                self._t_submit.pop(seq_id, None)
                return True
        if seq_id in self.swapped:
            self.decoder.pool.drop_swapped(seq_id)
            del self.swapped[seq_id]
            self._forget(seq_id)
            return True
        if seq_id in self.active:
            self._finish(seq_id)
            return True
        return False

    # ---- leaving -------------------------------------------------------------------------------------------
    def _forget(self, seq_id: int) -> None:
        self.active.pop(seq_id, None)
        self.last_logits.pop(seq_id, None)
        self.sequence_lengths.pop(seq_id, None)
        self._lb_pos.pop(seq_id, None)
        self._t_last.pop(seq_id, None)

    def _finish(self, seq_id: int) -> None:                  # remove_sequence_from_processing, :132-138
        self.decoder.pool.free(seq_id)
        self._forget(seq_id)

    def _keep_logits(self, seq_id: int) -> None:
        """A sequence that leaves the running batch alive keeps its row of the last decode call (a copy: a replayed
        hipGraph overwrites that tensor)."""
        pos = self._lb_pos.pop(seq_id, None)
        if pos is not None and seq_id not in self.last_logits:
            self.last_logits[seq_id] = self._lb_logits[pos].clone()

    def _preempt(self, victim: int) -> None:
        """handle_out_of_memory's victim (:117-130): swapped out to the host pool, or dropped."""
        pool = self.decoder.pool
        if self.preempt == "swap":
            try:
                self._keep_logits(victim)
                pool.swap_out(victim)
                self.swapped[victim] = self.active.pop(victim)
                self.stats["preemptions"] += 1
                return
            except RuntimeError as e:
                if "host blocks" not in str(e):
                    raise
        self._finish(victim)
        self.evicted.append(victim)
        self.stats["dropped"] += 1

    def _make_room(self, exclude: Sequence[int]) -> bool:
        """Preempt the most recently arrived sequence not in `exclude`, falling back to the youngest overall."""
        if not self.active:
            return False
        cands = [s for s in self.active if s not in exclude] or list(self.active)
        self._preempt(max(cands, key=self.active.get))
        return True

    # kept under its old name: tests and callers of round 5
    def _evict_youngest(self, exclude: Sequence[int]) -> bool:
        return self._make_room(exclude)

    # ---- coming (back) in ----------------------------------------------------------------------------------
    def _headroom(self, extra_running: int = 0) -> int:
        """Blocks the running sequences may ask for before a newcomer's first block boundary: one per layer each."""
        pool = self.decoder.pool
        per = getattr(pool, "num_layers", 1) if self.headroom_blocks is None else self.headroom_blocks
        return per * (len(self.active) + extra_running)

    def _resume(self) -> None:
        """Swapped-out sequences come back oldest first, while a batch slot is free and the pool holds their blocks plus
        the headroom of everything running (without it the newcomer would be the next victim at the next block boundary)."""
        pool = self.decoder.pool
        for sid in sorted(self.swapped, key=self.swapped.get):
            if len(self.active) >= self.max_batch:
                break
            need = pool.blocks_of(sid)
            forced = not self.active          # nothing runs: the oldest swapped sequence must, whatever the headroom
            if len(pool.free_blocks) < need + (0 if forced else self._headroom(1)):
                break                         # strictly oldest first: a younger, smaller sequence does not overtake
            if not pool.swap_in(sid):
                if forced:
                    raise RuntimeError(f"sequence {sid} cannot be swapped back in: {need} blocks needed, "
                                       f"{len(pool.free_blocks)} free, nothing else running")
                break
            self.active[sid] = self.swapped.pop(sid)
            self.stats["resumes"] += 1

    def _admit(self) -> None:
        """Queued requests are prefilled while a batch slot is free, nothing older waits on the host, and the pool holds
        their prompt's blocks plus the running sequences' headroom — as many as fit, through ONE prefill call."""
        if not self.waiting or self.swapped:
            return
        if self.active and self.steps - self._last_admit_step < self.admit_every:
            return
        pool = self.decoder.pool
        L, bs = getattr(pool, "num_layers", 1), getattr(pool, "block_size", 16)
        free = len(pool.free_blocks) - self._headroom()
        group, tokens = [], 0
        while self.waiting and len(self.active) + len(group) < self.max_batch:
            sid, ids = self.waiting[0]
            need = L * (-(-len(ids) // bs)) + L
            if need > free or (group and tokens + len(ids) > self.max_prefill_tokens):
                break
            self.waiting.popleft()
            group.append((sid, ids))
            free -= need
            tokens += len(ids)
        alone = False
        if not group and not self.active and self.waiting:      # nothing runs and the head of the queue does not "fit": try it alone
            group.append(self.waiting.popleft())
            alone = True
        if not group:
            return
        self._last_admit_step = self.steps
        try:
            self._prefill_group(group)
        except RuntimeError as e:
            # a request the EMPTY pool cannot hold (more blocks than exist, or a prompt past the block table) can never run: it is
            # refused — recorded with the reason, its id keeps answering sequences[...] with the prompt — and the loop goes on
            if not alone or not any(k in str(e) for k in ("free blocks", "blocks per layer", "single block per layer")):
                raise
            sid, ids = group[0]
            self.rejected[sid] = str(e)
            self.stop_at.pop(sid, None)
            self._t_submit.pop(sid, None)

    def _prefill_group(self, group) -> None:
        if hasattr(self.decoder, "prefill_batch") and (len(group) > 1 or getattr(self.decoder, "paged_prefill", False)):
            logits = self.decoder.prefill_batch([g[0] for g in group], [g[1] for g in group])
            self.stats["prefill_calls"] += 1
            for i, (sid, ids) in enumerate(group):
                self._started(sid, ids, logits[i])
        else:
            for sid, ids in group:
                self._started(sid, ids, self.decoder.prefill(sid, ids))
                self.stats["prefill_calls"] += 1

    # ---- one decode step for every running sequence ------------------------------------------------------
    def _batch_logits(self, batch: List[int]) -> torch.Tensor:
        if batch == self._lb_ids and self._lb_logits is not None and not any(s in self.last_logits for s in batch):
            return self._lb_logits
        # the members of the last call come out of its tensor with ONE gather; only newcomers (prefilled, resumed) are single rows
        pos = [-1 if s in self.last_logits else self._lb_pos.get(s, -1) for s in batch]
        new = [i for i, p in enumerate(pos) if p < 0]
        if len(new) == len(batch):
            return torch.stack([self.last_logits[s] for s in batch])
        lb = self._lb_logits
        if not new:
            return lb.index_select(0, self._index(pos, lb.device))
        # newcomers' rows ride at the end of the gather's source: one index_select over [last call's rows ; newcomers' rows]
        src = torch.cat([lb, torch.stack([self.last_logits[batch[i]] for i in new]).to(lb.dtype)])
        n0, k = lb.shape[0], 0
        for i in new:
            pos[i] = n0 + k
            k += 1
        return src.index_select(0, self._index(pos, lb.device))

    def _index(self, pos: List[int], device) -> torch.Tensor:
        """A row-index vector on `device`; on a GPU through a pinned buffer (torch.tensor(list, device=) stages synchronously:
        120 us a step).  One buffer is enough: step() waits for this step's ids, queued behind the copy, before the next call."""
        if device.type != "cuda":
            return torch.tensor(pos, dtype=torch.long)
        if getattr(self, "_idx_pin", None) is None or self._idx_pin.numel() < len(pos):
            self._idx_pin = torch.empty(max(len(pos), 2 * self.max_batch), dtype=torch.int64, pin_memory=True)
            self._idx_np = self._idx_pin.numpy()
        self._idx_np[: len(pos)] = pos
        return self._idx_pin[: len(pos)].to(device, non_blocking=True)

    def _tokens_to_host(self, tokens: torch.Tensor):
        """The sampled ids on their way to the host, requested NOW — ahead of the decode launches — and read after them."""
        if not tokens.is_cuda:
            return tokens, None
        if getattr(self, "_tok_pin", None) is None or self._tok_pin.numel() < tokens.numel():
            self._tok_pin = torch.empty(max(tokens.numel(), self.max_batch), dtype=torch.int64, pin_memory=True)
        host = self._tok_pin[: tokens.numel()]
        host.copy_(tokens, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    def step(self) -> List[int]:
        """Resume / admit what fits, then advance up to max_batch running sequences by one token; returns the ids stepped."""
        t0 = time.perf_counter()
        # sequences already at their limit end without sampling (:71-74)
        for sid in [s for s in self.active if self.sequence_lengths[s] >= self.stop_at[s]]:
            self._finish(sid)
        if self.swapped:
            self._resume()
        if self.waiting:
            ta = time.perf_counter()
            self._admit()
            ta = time.perf_counter() - ta
            self.stats["admit_s"] += ta
            t0 += ta                                   # (prefill calls are accounted apart from the decode step's host time)
        batch = sorted(self.active, key=self.active.get)[: self.max_batch]      # oldest first (PriorityQueue, :16)
        if not batch:
            return []
        for s in self._lb_ids:               # members of the last call that sit this one out keep their logits
            if s not in self.last_logits and s in self._lb_pos and s in self.active and s not in batch:
                self._keep_logits(s)
        logits = self._batch_logits(batch)
        tokens = self.sampler(logits, generator=self.generator)                 # :76
        tok_host, tok_ev = self._tokens_to_host(tokens)
        keep = None
        while True:
            try:
                new_logits = self.decoder.decode(batch, tokens, use_graph=self.use_graph) \
                    if self.use_graph else self.decoder.decode(batch, tokens)
                break
            except RuntimeError as e:                                           # :110-115
                if "free blocks" not in str(e):
                    raise
                # the reference picks the youngest sequence other than the one being processed; with a
                # batch in flight, prefer a victim outside the batch, else the youngest member
                outside = [s for s in self.active if s not in batch]
                victim = max(outside or batch, key=self.active.get)
                if victim in batch:          # its logits row must survive: it samples again when it is back
                    i = batch.index(victim)
                    if victim not in self.last_logits:
                        self.last_logits[victim] = logits[i if keep is None else keep[i]].clone()
                    self._lb_pos.pop(victim, None)
                self._preempt(victim)
                if victim in batch:
                    sel = [j for j, s in enumerate(batch) if s != victim]
                    keep = sel if keep is None else [keep[j] for j in sel]
                    batch = [batch[j] for j in sel]
                    tokens = tokens[sel]
                if not batch:
                    return []
                # NB: a step that failed mid-way may have advanced some rows' bookkeeping; the pool hands
                # out blocks before touching fill counters, so retrying is consistent (kv_pool._step_rows)
        t1 = time.perf_counter()
        if tok_ev is not None:
            tok_ev.synchronize()             # (the ids left the GPU ahead of this step's layers: no wait for those)
        tok_list = tok_host.tolist()
        if keep is not None:
            tok_list = [tok_list[j] for j in keep]
        t2 = time.perf_counter()
        for s in batch:
            self.last_logits.pop(s, None)
        self._lb_ids, self._lb_logits = list(batch), new_logits
        self._lb_pos = {s: i for i, s in enumerate(batch)}
        for i, sid in enumerate(batch):
            self.sequences[sid].append(tok_list[i])                             # :79
            n = self.sequence_lengths[sid] = self.sequence_lengths[sid] + 1     # :101
            if tok_list[i] == self.eos_token_id or n >= self.stop_at[sid]:      # :103-108
                self._finish(sid)
        self.steps += 1
        self.stats["decode_rows"] += len(batch)
        if self.record_latency:
            now = time.perf_counter()
            lat = np.fromiter((now - self._t_last.get(s, self._t_submit.get(s, now)) for s in batch), dtype=np.float64,
                              count=len(batch))
            for s in batch:
                if s not in self._t_last and s in self._t_submit:
                    self.first_token_s.append(now - self._t_submit.pop(s))
                if s in self.active:
                    self._t_last[s] = now
            self.token_latency_s.append(lat)
        t3 = time.perf_counter()
        self.stats["host_s"] += (t1 - t0) + (t3 - t2)
        self.stats["wait_s"] += t2 - t1
        return batch

    def pending(self) -> bool:
        return bool(self.active or self.swapped or self.waiting)

    def run(self, max_steps: Optional[int] = None) -> int:
        """Step until no sequence is running, swapped out or queued (scheduler.py:55-115); returns the number of steps."""
        n = 0
        while self.pending() and (max_steps is None or n < max_steps):
            self.step()
            n += 1
        return n


def deal_requests(num_requests: int, rank: int, world: int) -> List[int]:
    """Arrival index -> rank, round robin (keeps each rank's batch the same age mix)."""
    return [i for i in range(num_requests) if i % world == rank]
