# Builds the C-ABI libraries (include/vmi_paged_attention.h; the extras library also include/vmi_paged_attention_extras.h) without Python: `make -j` — the same commands as
# `python -m vllmini_amd.build [--extras] [--diag]` (vllmini_amd/build.py), one object per translation unit.  gfx950 only.
#   make            the PRODUCT library (what the operators load: the hot path of SURVEY.md §8 — float16 tensors over
#                   float16 / fp8-E4M3 pages; the out-of-scope kernel menus are empty, pa_extras_absent.hip) + the CPU oracle
#   make extras     libvmi_paged_attention_extras.so too: the product's objects + bfloat16 / float32 tensors, fp8-E5M2 pages,
#                   block-sparse attention, reshape_and_cache_flash, convert_fp8 (opt-in: vllmini_amd._lib.use_extras())
#   make diag       ... and the diagnostic library (the extras library under -DVMI_DIAG: include/vmi_paged_attention_diag.h's
#                   entries, the "loads only" variants, the LDS-staging experiment); it shares every object -DVMI_DIAG
#                   does not change
#   make layer      libvmi_gpt2_layer.so (include/vmi_gpt2_layer.h): the decode harness's own library — the GPT-2 block's
#                   linear layers; one unit, part of `make` / `all`
HIPCC  ?= hipcc
ARCH   ?= gfx950
# -ffp-contract=off: the fp16 p*v products are rounded before the fp16 adds (the reference's rounding points)
FLAGS  := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -fno-gpu-rdc -Iinclude
CSRC   := vllmini_amd/csrc
OUTDIR := vllmini_amd/_C
CORE_UNITS := paged_attention pa_variants_extra pa_append_core pa_append_extra pa_variants_fp8 pa_queue pa_split
ABSENT_UNITS := pa_extras_absent
EXTRAS_UNITS := pa_variants_bf16 pa_append_bf16 pa_variants_fp8_bf16 pa_variants_fp8_e5m2 pa_variants_fp8_e5m2_bf16 \
          pa_variants_sparse pa_variants_sparse_bf16 pa_f32 pa_extras_cache pa_extras_abi
DIAG_UNITS := paged_attention pa_append_core pa_variants_fp8 pa_queue pa_split pa_stage
# pa_queue.hip holds its bfloat16 / E5M2 rows behind -DVMI_EXTRAS: one object for the product, one for the other two
OBJS   := $(CORE_UNITS:%=$(OUTDIR)/%.hip.o) $(ABSENT_UNITS:%=$(OUTDIR)/%.hip.o)
EXTRAS_OBJS := $(filter-out $(OUTDIR)/pa_queue.hip.o,$(CORE_UNITS:%=$(OUTDIR)/%.hip.o)) $(OUTDIR)/pa_queue.hip.extras.o \
          $(EXTRAS_UNITS:%=$(OUTDIR)/%.hip.o)
DIAG_OBJS := $(DIAG_UNITS:%=$(OUTDIR)/%.hip.diag.o) $(filter-out $(DIAG_UNITS:%=$(OUTDIR)/%.hip.o) $(OUTDIR)/pa_queue.hip.extras.o,$(EXTRAS_OBJS))
LIB    := $(OUTDIR)/libvmi_paged_attention.so
EXTRAS_LIB := $(OUTDIR)/libvmi_paged_attention_extras.so
DIAG_LIB := $(OUTDIR)/libvmi_paged_attention_diag.so
DEPS   := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inc) include/vmi_paged_attention.h include/vmi_paged_attention_extras.h

LAYER_LIB := $(OUTDIR)/libvmi_gpt2_layer.so

all: $(LIB) $(LAYER_LIB) oracle

layer: $(LAYER_LIB)

$(LAYER_LIB): $(CSRC)/gpt2_layer.hip include/vmi_gpt2_layer.h
	mkdir -p $(OUTDIR)
	$(HIPCC) $(filter-out -ffp-contract=off,$(FLAGS)) -shared $< -o $@

extras: $(LIB) $(EXTRAS_LIB)

diag: $(LIB) $(EXTRAS_LIB) $(DIAG_LIB)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fno-gpu-rdc $^ -o $@

$(EXTRAS_LIB): $(EXTRAS_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fno-gpu-rdc $^ -o $@

$(DIAG_LIB): $(DIAG_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fno-gpu-rdc $^ -o $@

$(OUTDIR)/%.hip.o: $(CSRC)/%.hip $(DEPS)
	mkdir -p $(OUTDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OUTDIR)/%.hip.extras.o: $(CSRC)/%.hip $(DEPS)
	mkdir -p $(OUTDIR)
	$(HIPCC) $(FLAGS) -DVMI_EXTRAS -c $< -o $@

$(OUTDIR)/%.hip.diag.o: $(CSRC)/%.hip $(DEPS) include/vmi_paged_attention_diag.h
	mkdir -p $(OUTDIR)
	$(HIPCC) $(FLAGS) -DVMI_DIAG -DVMI_EXTRAS -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(OUTDIR) oracle/_build

.PHONY: all extras diag oracle clean
