// dtof_shade_res1.hip -- instantiations of k_shade (dtof_shade.h): the resident first-bounce kernel, every BSDF / emitter / texture.
#include "dtof_shade.h"

namespace dtof {

void launch_shade_resident1(bool k4, const ShadeLaunch &L) {
    if (k4) launch_resident_variant<true, kMaxOffsets, 1>(L); else launch_resident_variant<true, 1, 1>(L);
}

}  // namespace dtof
