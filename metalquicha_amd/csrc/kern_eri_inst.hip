// kern_eri_inst.hip -- explicit instantiations of the ERI class kernels, one group per object file
// (build.sh compiles this file once per ERI_GROUP so the 21 classes build in parallel).
#include "eri_kernels.hpp"

namespace mqc {
#define ERI_INST(a, b, c, d) \
    template void launch_eri_class<a, b, c, d>(const BatchView&, const int*, int, const int*, int, const double*, double, hipStream_t);
#define DIG_INST(a, b, c, d) \
    template void launch_eri_digest_class<a, b, c, d>(const BatchView&, const int*, int, const double*, const double*, double, double*, double*, int, hipStream_t);
#define TWIN_INST(a, b, c, d) \
    template void launch_eri_twin_class<a, b, c, d>(const BatchView&, const int*, int, const int*, int, const double*, double, hipStream_t);
#define TWINW_INST(a, b, c, d) \
    template void launch_eri_twin_wave_class<a, b, c, d>(const BatchView&, const int*, int, const double*, double, hipStream_t);
#define SCHWARZ_INST(a, b) \
    template void launch_schwarz_class<a, b>(const BatchView&, const int*, int, int*, double*, hipStream_t);

#if ERI_GROUP == 0
ERI_INST(0, 0, 0, 0) ERI_INST(1, 0, 0, 0) ERI_INST(1, 0, 1, 0) ERI_INST(1, 1, 0, 0) ERI_INST(1, 1, 1, 0) ERI_INST(1, 1, 1, 1)
ERI_INST(2, 0, 0, 0) ERI_INST(2, 0, 1, 0) ERI_INST(2, 0, 1, 1) ERI_INST(2, 0, 2, 0)
#elif ERI_GROUP == 1
ERI_INST(2, 1, 0, 0) ERI_INST(2, 1, 1, 0) ERI_INST(2, 1, 1, 1)
#elif ERI_GROUP == 2
ERI_INST(2, 1, 2, 0) ERI_INST(2, 1, 2, 1)
#elif ERI_GROUP == 3
ERI_INST(2, 2, 0, 0) ERI_INST(2, 2, 1, 0)
#elif ERI_GROUP == 4
ERI_INST(2, 2, 1, 1)
#elif ERI_GROUP == 5
ERI_INST(2, 2, 2, 0)
#elif ERI_GROUP == 6
ERI_INST(2, 2, 2, 1)
#elif ERI_GROUP == 15
ERI_INST(2, 2, 2, 2)
#elif ERI_GROUP == 7
SCHWARZ_INST(0, 0) SCHWARZ_INST(1, 0) SCHWARZ_INST(1, 1) SCHWARZ_INST(2, 0)
#elif ERI_GROUP == 18
SCHWARZ_INST(2, 2)
#elif ERI_GROUP == 19
SCHWARZ_INST(2, 1)
#elif ERI_GROUP == 8
DIG_INST(0, 0, 0, 0) DIG_INST(1, 0, 0, 0) DIG_INST(1, 0, 1, 0) DIG_INST(1, 1, 0, 0) DIG_INST(1, 1, 1, 0) DIG_INST(1, 1, 1, 1)
DIG_INST(2, 0, 0, 0) DIG_INST(2, 0, 1, 0) DIG_INST(2, 0, 1, 1) DIG_INST(2, 0, 2, 0)
#elif ERI_GROUP == 9
DIG_INST(2, 1, 0, 0) DIG_INST(2, 1, 1, 0) DIG_INST(2, 1, 1, 1)
#elif ERI_GROUP == 10
DIG_INST(2, 1, 2, 0) DIG_INST(2, 1, 2, 1)
#elif ERI_GROUP == 11
DIG_INST(2, 2, 0, 0) DIG_INST(2, 2, 1, 0)
#elif ERI_GROUP == 12
DIG_INST(2, 2, 1, 1)
#elif ERI_GROUP == 13
DIG_INST(2, 2, 2, 0)
#elif ERI_GROUP == 16
TWIN_INST(0, 0, 0, 0) TWIN_INST(1, 0, 0, 0) TWIN_INST(1, 0, 1, 0) TWIN_INST(1, 1, 0, 0)
TWINW_INST(0, 0, 0, 0) TWINW_INST(1, 0, 0, 0)
#elif ERI_GROUP == 17
TWIN_INST(1, 1, 1, 0) TWIN_INST(2, 0, 0, 0) TWIN_INST(2, 0, 1, 0) TWIN_INST(2, 1, 0, 0)
#elif ERI_GROUP == 14
DIG_INST(2, 2, 2, 1) DIG_INST(2, 2, 2, 2)
#endif
}  // namespace mqc
