// cp_fftlog_inst.hip -- kernel instantiations for one size group (compiled once per CP_INST_GROUP so the
// groups build in parallel; see Makefile).
#include "cp_fftlog_kernel.h"

#ifndef CP_INST_GROUP
#error "compile with -DCP_INST_GROUP=0..4"
#endif

namespace cpfft {

#define CP_CAT2(a, b) a##b
#define CP_CAT(a, b) CP_CAT2(a, b)

#if CP_INST_GROUP == 0
#define CP_IN_GROUP(NP_) (NP_ <= 256)
#elif CP_INST_GROUP == 1
#define CP_IN_GROUP(NP_) (NP_ == 512 || NP_ == 1024)
#elif CP_INST_GROUP == 2
#define CP_IN_GROUP(NP_) (NP_ == 2048)
#elif CP_INST_GROUP == 3
#define CP_IN_GROUP(NP_) (NP_ == 4096)
#else
#define CP_IN_GROUP(NP_) (NP_ == 8192)
#endif

bool CP_CAT(find_launcher_g, CP_INST_GROUP)(int npad, Launcher* out) {
#define X(NP_, P_)                                   \
    if constexpr (CP_IN_GROUP(NP_)) {                \
        if (npad == NP_) {                           \
            *out = make_launcher<NP_, P_>();         \
            return true;                             \
        }                                            \
    }
    CP_FFTLOG_SIZES(X)
#undef X
    return false;
}

}  // namespace cpfft
