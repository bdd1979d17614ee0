// dtof_shade_spec1.hip -- instantiations of k_shade (dtof_shade.h): every BSDF / emitter / texture (SPEC = 1).
#include "dtof_shade.h"

namespace dtof {

void launch_shade_spec1(bool k4, const ShadeLaunch &L) {
    if (k4) launch_shade_variant<true, kMaxOffsets, true, 1>(L); else launch_shade_variant<true, 1, true, 1>(L);
}

}  // namespace dtof
