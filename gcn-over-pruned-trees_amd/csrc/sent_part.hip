// One (precision combination, direction) slice of the sentence-slice kernel's instantiations: compiled 15 times with
// -DGCNPT_SS_PART=0..14 (combo = part / 3, mode = part % 3; see sent_body.h / sent_common.h) so that the build is parallel.
#include "sent_common.h"

namespace gcnpt {

template <typename CT, typename IT, typename OT, int MODE, int VEC, int CFG>
static int launch_cfg(hipStream_t s, const SentParams& p, size_t lds, int grid) {
    auto kern = sent_kernel<CT, IT, OT, MODE, VEC, ss_ksh(CFG)>;
    GCNPT_LDS_ATTR_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SS_THREADS), lds, s, p);
    note_launch(grid, SS_THREADS, lds, sizeof(p));
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
}

template <typename CT, typename IT, typename OT, int MODE>
static int launch_mode(hipStream_t s, const SentParams& p, int cfg, int vec, size_t lds, int grid) {
    // fp32 rows feeding bf16 fragments and the dZ-deriving loader keep twice the registers per k-step: plan_sent gives them the
    // 4-k-step configuration only, the wider ones are not built
    constexpr bool HEAVY = (sizeof(CT) == 2 && sizeof(IT) == 4) || MODE == 1;
    if constexpr (HEAVY) {
        if (cfg != 0) return fail(GCNPT_E_UNSUPPORTED, "sentence-slice layer: configuration %d is not built for this precision", cfg);
        if (vec == 8) return launch_cfg<CT, IT, OT, MODE, 8, 0>(s, p, lds, grid);
        return launch_cfg<CT, IT, OT, MODE, 4, 0>(s, p, lds, grid);
    } else {
        if (vec == 8) {
            if (cfg == 0) return launch_cfg<CT, IT, OT, MODE, 8, 0>(s, p, lds, grid);
            if (cfg == 1) return launch_cfg<CT, IT, OT, MODE, 8, 1>(s, p, lds, grid);
            return launch_cfg<CT, IT, OT, MODE, 8, 2>(s, p, lds, grid);
        }
        if (cfg == 0) return launch_cfg<CT, IT, OT, MODE, 4, 0>(s, p, lds, grid);
        if (cfg == 1) return launch_cfg<CT, IT, OT, MODE, 4, 1>(s, p, lds, grid);
        return launch_cfg<CT, IT, OT, MODE, 4, 2>(s, p, lds, grid);
    }
}

#define GCNPT_SS_NAME2(n) sent_launch_part##n
#define GCNPT_SS_NAME(n) GCNPT_SS_NAME2(n)
int GCNPT_SS_NAME(GCNPT_SS_PART)(hipStream_t s, const SentParams& p, int cfg, int vec, size_t lds, int grid) {
    constexpr int COMBO = GCNPT_SS_PART / 3, MODE = GCNPT_SS_PART % 3;
    if constexpr (COMBO == 0) return launch_mode<float, float, float, MODE>(s, p, cfg, vec, lds, grid);
    else if constexpr (COMBO == 1) return launch_mode<bf16_t, float, float, MODE>(s, p, cfg, vec, lds, grid);
    else if constexpr (COMBO == 2) return launch_mode<bf16_t, float, bf16_t, MODE>(s, p, cfg, vec, lds, grid);
    else if constexpr (COMBO == 3) return launch_mode<bf16_t, bf16_t, float, MODE>(s, p, cfg, vec, lds, grid);
    else return launch_mode<bf16_t, bf16_t, bf16_t, MODE>(s, p, cfg, vec, lds, grid);
}

}  // namespace gcnpt
