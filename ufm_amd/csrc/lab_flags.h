// The lab flag words of libufm_hip.so (ufm_debug_set_gemm_flags, ufm_debug_set_conv_variant): ONE table of {name, shift, width}
// per word.  Every consumer -- host dispatch and device code alike -- reads its field through lab_get(), which masks to the
// field's own width; the tables are checked at compile time to be pairwise disjoint; the setters refuse any bit outside the table
// (UFM_ERR_ARG).  Why: until late in round 5 the 8-phase GEMM decoded its rasterization height as `flags >> 8`, so every lab bit
// added above bit 15 silently switched one arm of four A/Bs to column-major order.  With this file that class of error does not compile
// (overlap) or is refused at the call (unknown bit).  tests/test_abi_cpu.py walks the tables through ufm_debug_lab_field().
#pragma once

struct LabField {
    const char* name;
    int shift, width;
};

constexpr unsigned lab_mask(const LabField& f) { return ((f.width >= 32 ? 0u : (1u << f.width)) - 1u) << f.shift; }
constexpr int lab_get(int word, const LabField& f) { return (int)(((unsigned)word >> f.shift) & ((1u << f.width) - 1u)); }
template <int N>
constexpr bool lab_disjoint(const LabField (&t)[N]) {
    unsigned seen = 0;
    for (int i = 0; i < N; ++i) {
        if (t[i].width < 1 || t[i].shift < 0 || t[i].shift + t[i].width > 31) return false;  // bit 31 stays clear: the words travel as int
        if (seen & lab_mask(t[i])) return false;
        seen |= lab_mask(t[i]);
    }
    return true;
}
template <int N>
constexpr unsigned lab_known(const LabField (&t)[N]) {
    unsigned m = 0;
    for (int i = 0; i < N; ++i) m |= lab_mask(t[i]);
    return m;
}

// ---- word 0: ufm_debug_set_gemm_flags (GemmArgs::debug on the device) ----
namespace gemm_lab {
constexpr LabField NO_DMA{"no_dma", 1, 1};                // timing ablation: no LDS-DMA (results wrong)
constexpr LabField NO_EPILOGUE{"no_epilogue", 2, 1};      // timing ablation: no epilogue traffic (results wrong)
constexpr LabField DIRECT_EPILOGUE{"direct_epilogue", 3, 1};  // the un-staged epilogue of the 128x128 kernel (A/B)
constexpr LabField LDA0{"lda0", 4, 1};                    // lda = 0: every tile reads the same rows (all-L2-hit probe)
constexpr LabField LDW0{"ldw0", 5, 1};                    // ldw = 0
constexpr LabField GENERIC_EPILOGUE{"generic_epilogue", 6, 1};  // run-time switched epilogue in the 8-phase kernels
constexpr LabField NO_TWO_KTILES{"no_two_ktiles", 7, 1};  // never the two-K-tiles-per-barrier form of the 128x128 kernel
constexpr LabField RASTER_GROUP{"raster_group", 8, 8};    // grouped-rasterization height of the 8-phase kernel (0 = 8)
constexpr LabField STAGGER{"stagger", 16, 7};             // first-round start stagger (pair kernel: all 7 bits; 8-phase kernel: its low 3)
constexpr LabField SERIAL_RMW{"serial_rmw", 23, 1};       // the serial read-modify-write read-out of rounds 1-4 (A/B)
constexpr LabField PAIR_FLIP{"pair_flip", 24, 4};         // flip the auto dispatch's four pair-kernel rules (A/B)
constexpr LabField NO_PERSIST{"no_persist", 28, 1};       // never the persistent 8-phase kernel in auto (A/B)
constexpr LabField LATENCY{"latency_objective", 29, 1};   // tile heights for the launch's own latency on every stream
constexpr LabField CU_TIME{"cu_time_objective", 30, 1};   // tile heights for CU time on every stream
constexpr LabField ALL[] = {NO_DMA, NO_EPILOGUE, DIRECT_EPILOGUE, LDA0, LDW0, GENERIC_EPILOGUE, NO_TWO_KTILES, RASTER_GROUP,
                            STAGGER, SERIAL_RMW, PAIR_FLIP, NO_PERSIST, LATENCY, CU_TIME};
static_assert(lab_disjoint(ALL), "gemm lab flag fields overlap");
}  // namespace gemm_lab

// ---- word 1: ufm_debug_set_conv_variant ----
namespace conv_lab {
constexpr LabField KERNEL{"kernel", 0, 4};                // 0 auto, 1 = 128-row kernels, 2 = 8-phase wherever applicable, 3 = 128-row and never the deep ring, 4 = pair kernel wherever applicable
constexpr LabField SERIAL_EPILOGUE{"serial_epilogue", 4, 1};  // "+ 16": the per-pass residual read-out of rounds 1-4 (A/B)
constexpr LabField HALO{"halo", 5, 2};                    // round 6: 0 = the row-window halo form of the 8-phase 3x3 kernel wherever it applies, 1 = never (the gather form: A/B, tests), 2 = the halo form with planar W staging even where an interleaved copy is registered (A/B)
constexpr LabField NF_PIN{"nf_pin", 8, 4};                // pinned 8-phase tile height (5..8 fragments per wave row)
constexpr LabField ABLATE{"ablate", 12, 7};               // timing ablations of the stamped 8-phase loop (ConvX3Args::ablate)
constexpr LabField LATENCY{"latency_objective", 19, 1};
constexpr LabField CU_TIME{"cu_time_objective", 20, 1};
constexpr LabField ALL[] = {KERNEL, SERIAL_EPILOGUE, HALO, NF_PIN, ABLATE, LATENCY, CU_TIME};
static_assert(lab_disjoint(ALL), "conv lab flag fields overlap");
}  // namespace conv_lab
