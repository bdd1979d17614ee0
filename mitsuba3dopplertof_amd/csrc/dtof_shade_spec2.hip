// dtof_shade_spec2.hip -- instantiations of k_shade (dtof_shade.h): every BSDF + blendbsdf / two-BSDF twosided (SPEC = 2: the BSDF chain loops over two records).
#include "dtof_shade.h"

namespace dtof {

void launch_shade_spec2(bool k4, const ShadeLaunch &L) {
    if (k4) launch_shade_variant<true, kMaxOffsets, true, 2>(L); else launch_shade_variant<true, 1, true, 2>(L);
}

}  // namespace dtof
