// Prints the LDS bytes per instance of the 4-wavefront kernel for every model build (host program: constexpr only, no GPU call).
//   hipcc -O0 --offload-arch=gfx950 -std=c++17 -Iinclude -Isrbd_horizon_amd/csrc tools/lds_budget.hip -o build/lds_budget
// tests/test_lds_budget.py pins what DESIGN.md section 5 claims (two workgroups per CU for srbd37 / lip30).
#include "sddp_kernels.hpp"
#include "sddp_kernels_mw.hpp"
#include <cstdio>
using namespace sddp;
template <class M> void show(const char* n) {
    using L = LdsMW<M>;
    std::printf("%s %zu %d %d\n", n, L::BYTES, L::WORK, int(2 * L::BYTES <= size_t(160) * 1024));
}
int main() {
    show<Srbd37>("srbd37"); show<Srbd37B>("srbd37B"); show<Srbd37S>("srbd37S"); show<Srbd37BS>("srbd37BS"); show<Lip30>("lip30"); show<Srbd61>("srbd61"); show<Srbd37X>("srbd37X"); show<Lip30X>("lip30X"); show<Srbd61X>("srbd61X"); show<Srbd61B>("srbd61B");
    std::printf("srbd13 %zu %d %d\n", Lds<Srbd13>::BYTES, 0, 1);
    return 0;
}
