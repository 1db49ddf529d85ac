// pa_variants_fp8.hip — paged_attention_v1 / v2-partition instantiations for an fp8 E4M3 KV cache (pa_table_fp8.inc;
// kv_cache_dtype "fp8" / "fp8_e4m3").  SURVEY.md section 8 row f-4.  The E5M2 twin is pa_variants_fp8_e5m2.hip.
#define VMI_F8_FMT 1
#define VMI_F8_PFX "fp8_"
#define VMI_F8_SYM(x) x
#include "pa_kernel.hpp"
#include "pa_variants_fp8_body.inc"
