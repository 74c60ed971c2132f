// One (precision combination, direction) slice of the row-tile kernel's instantiations: compiled 15 times with -DGCNPT_RT_PART=0..14
// (combo = part / 3, mode = part % 3; see rowtile_body.h) so that the build is parallel.
#include "colsplit_body.h"

namespace gcnpt {

template <typename CT, typename IT, typename OT>
static int launch_mode(hipStream_t s, const RowTileParams& p) {
    constexpr int MODE = GCNPT_RT_PART % 3;
    if constexpr (MODE == 0) return launch_rowtile<CT, IT, OT, false, false>(s, p);
    else if constexpr (MODE == 1) return launch_rowtile<CT, IT, OT, true, false>(s, p);
    else return launch_rowtile<CT, IT, OT, true, true>(s, p);
}

#define GCNPT_RT_NAME2(n) rowtile_launch_part##n
#define GCNPT_RT_NAME(n) GCNPT_RT_NAME2(n)
int GCNPT_RT_NAME(GCNPT_RT_PART)(hipStream_t s, const RowTileParams& p) {
    constexpr int COMBO = GCNPT_RT_PART / 3;
    if constexpr (COMBO == 0) return launch_mode<float, float, float>(s, p);
    else if constexpr (COMBO == 1) return launch_mode<bf16_t, float, float>(s, p);
    else if constexpr (COMBO == 2) return launch_mode<bf16_t, float, bf16_t>(s, p);
    else if constexpr (COMBO == 3) return launch_mode<bf16_t, bf16_t, float>(s, p);
    else return launch_mode<bf16_t, bf16_t, bf16_t>(s, p);
}

}  // namespace gcnpt
