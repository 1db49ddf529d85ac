# Builds the C-ABI libraries (include/vmi_paged_attention.h) without Python: `make -j` — the same commands as
# `python -m vllmini_amd.build [--diag]` (vllmini_amd/build.py), one object per translation unit.  gfx950 only.
#   make            the product library (what the operators load) + the CPU oracle
#   make diag       the diagnostic library too (-DVMI_DIAG: include/vmi_paged_attention_diag.h's entries, the "loads
#                   only" variants, the LDS-staging experiment); it shares every object -DVMI_DIAG does not change
HIPCC  ?= hipcc
ARCH   ?= gfx950
# -ffp-contract=off: the fp16 p*v products are rounded before the fp16 adds (the reference's rounding points)
FLAGS  := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc -Iinclude
CSRC   := vllmini_amd/csrc
OUTDIR := vllmini_amd/_C
UNITS  := paged_attention pa_variants_extra pa_variants_bf16 pa_append_core pa_append_extra pa_append_bf16 \
          pa_variants_fp8 pa_variants_fp8_bf16 pa_variants_fp8_e5m2 pa_variants_fp8_e5m2_bf16 \
          pa_variants_sparse pa_variants_sparse_bf16 pa_f32 pa_queue
DIAG_UNITS := paged_attention pa_append_core pa_stage
OBJS   := $(UNITS:%=$(OUTDIR)/%.hip.o)
DIAG_OBJS := $(DIAG_UNITS:%=$(OUTDIR)/%.hip.diag.o) $(filter-out $(DIAG_UNITS:%=$(OUTDIR)/%.hip.o),$(OBJS))
LIB    := $(OUTDIR)/libvmi_paged_attention.so
DIAG_LIB := $(OUTDIR)/libvmi_paged_attention_diag.so
DEPS   := $(CSRC)/pa_kernel.hpp $(CSRC)/pa_queue.hpp $(wildcard $(CSRC)/*.inc) include/vmi_paged_attention.h

all: $(LIB) oracle

diag: $(LIB) $(DIAG_LIB)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fno-gpu-rdc $^ -o $@

$(DIAG_LIB): $(DIAG_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fno-gpu-rdc $^ -o $@

$(OUTDIR)/%.hip.o: $(CSRC)/%.hip $(DEPS)
	mkdir -p $(OUTDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OUTDIR)/%.hip.diag.o: $(CSRC)/%.hip $(DEPS) include/vmi_paged_attention_diag.h
	mkdir -p $(OUTDIR)
	$(HIPCC) $(FLAGS) -DVMI_DIAG -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(OUTDIR) oracle/_build

.PHONY: all diag oracle clean
