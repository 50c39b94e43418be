// kernels_kpp.hpp -- KPP vertical mixing (placeholder until the KPP kernels land)
#pragma once
namespace pop {
inline int kpp_create(HostModel &, const DevGrid &, MixDev &, std::vector<void *> &, std::string &err) { err = "KPP vertical mixing is not built yet"; return 1; }
inline int kpp_vmix_coeffs(const HostModel &, const DevGrid &, const StepParams &, const MixDev &, const MixState &, hipStream_t, std::string &err) { err = "KPP not built"; return 1; }
}  // namespace pop
