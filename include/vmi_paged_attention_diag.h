/*
 * vmi_paged_attention_diag.h — entries that exist ONLY in the diagnostic build of the library
 * (vllmini_amd/_C/libvmi_paged_attention_diag.so: the product sources compiled with -DVMI_DIAG; `python -m
 * vllmini_amd.build --diag`).  None of them has a reference counterpart and none is exported by the product library
 * (tests/test_abi.py checks that with the dynamic symbol table).  The diagnostic build also carries kernels the product
 * does not: the "LOADSONLY" variants (the page gather with the math removed — wrong results by design) and the
 * LDS-staging experiment ("stage_*", vllmini_amd/csrc/pa_stage.hip).
 */
#ifndef VMI_PAGED_ATTENTION_DIAG_H
#define VMI_PAGED_ATTENTION_DIAG_H

#include "vmi_paged_attention.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the entries declared here are all it exports */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/*
 * Test / benchmark knob of the balanced ("q_*") kernels (per host thread, default 0 = automatic; returns the previous
 * value): forces their mode, worker count or hand-out policy (bit layout: vllmini_amd/csrc/pa_queue.hpp, QF_*).
 * Results do not depend on it — every mode computes an item with the same operations in the same order — only the
 * schedule does; tests use it to drive every path of the kernel on small inputs.
 */
int vmi_debug_set_queue_flags(int32_t flags);

/*
 * Diagnostic (no reference counterpart): plain coalesced 16-B/lane read of `bytes` from `src`
 * with `blocks` workgroups of 256 threads; nt != 0 uses non-temporal loads.  `sink` is a 4-byte
 * device word that is (practically) never written.  scripts/bench_diag.py --diag uses it to report the read
 * bandwidth this box sustains, next to the attention kernel's achieved figure.
 */
int vmi_diag_stream_read(const void* src, int64_t bytes, void* sink, int32_t blocks, int32_t nt,
                         int32_t device, void* stream);

/*
 * Diagnostic: read `bytes` from `src` as pseudo-randomly ordered contiguous chunks of chunk_kb KiB
 * (1..64, power of two), one chunk stream per wave, inflight_kb KiB (1,2,4,8,16) requested per wave before
 * anything is consumed — the attention kernel's access pattern without the math, with the contiguous-chunk
 * size and the queue depth as the variables.
 */
int vmi_diag_gather_read(const void* src, int64_t bytes, void* sink, int32_t chunk_kb, int32_t inflight_kb,
                         int32_t blocks, int32_t nt, int32_t device, void* stream);

/*
 * Diagnostic: a timeline of the balanced ("q_*") kernels' waves.  `records` is device memory for 4 x uint64 per wave of
 * the launch (grid x 4 waves, in workgroup order: {start, end} in ticks of the constant 100 MHz clock (s_memrealtime),
 * HW_REG_HW_ID, HW_REG_XCC_ID), written by every later launch of such a kernel on `device` until it is set to NULL again.
 * scripts/wave_timeline_probe.py reads it: when the waves of a launch finish, per XCD and CU.  Synchronous.
 */
int vmi_diag_set_wave_timeline(void* records, int32_t device);

/*
 * Diagnostic: stage stamps of pa_v1_kernel (the kernels of the block-16 x head-64/128 menu, pa_table_core.inc).
 * `records` is device memory for 12 x uint64 per wave of the launch, in (z, y, x) workgroup order x wave: ten stamps in ticks
 * of the constant 100 MHz clock — entry, seq_len known, first pages requested, first K group consumed, K pass done, maxima
 * exchanged, probabilities written, V pass done, partial outputs exchanged, out stored — then HW_REG_HW_ID and
 * HW_REG_XCC_ID | (blocks of this wave << 8).  Written by every later launch on `device` until set to NULL again.
 * scripts/stage_timeline_probe.py reads it: the latency chain of an under-filled chip, term by term.  Synchronous.
 */
int vmi_diag_set_stage_stamps(void* records, int32_t device);

/*
 * Diagnostic: stage stamps of the split kernels (vllmini_amd/csrc/pa_split.hpp).  `records` is device memory for 10 x uint64
 * per wave of the launch, in workgroup order x wave: eight stamps in ticks of the constant 100 MHz clock — entry, lengths
 * known, first K group consumed, K pass done, granule published, exchange complete, V pass done, end — then HW_REG_HW_ID
 * and HW_REG_XCC_ID | (blocks of this wave << 8).  Synchronous.
 */
int vmi_diag_set_split_stamps(void* records, int32_t device);
/* Test knob of the split kernels (per host thread; returns the previous value): bit 0 = an item's workgroups far apart
 * in dispatch order instead of adjacent. */
int vmi_debug_set_split_flags(int32_t flags);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VMI_PAGED_ATTENTION_DIAG_H */
