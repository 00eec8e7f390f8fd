// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_rough_driver.cpp — extern "C" driver around the reference's own RoughTransmittance::Evaluate / EvaluateDiffuse (Engine/RoughTransmittance.cu:55-121): the warp of
// (cos theta, alpha, eta) into the table's parameter space, the eta < 1 block, the clamps, and the call into Spline::evalCubicInterp3D / 2D — the lookup behind roughplastic and
// roughcoating — and the manager's lookup by distribution type (:123, :140-158).  `make ref` compiles exactly those line ranges through a build-time extract under oracle/_ref/gen/ (git-ignored) behind the reference's own headers
// Engine/RoughTransmittance.h and Math/Spline.h, with the same one-line preface as Math/Spline.cu (`using std::min; using std::max;`).  The file's other functions are left out:
// the constructor reads Mitsuba's .dat files through CUDA_MALLOC / cudaMemcpy, StaticInitialize fills the device copy with cudaMemcpyToSymbol.
// The class has no other way in than that constructor and its members are private, so the driver lays a table out in raw storage the size of the object, in the member order
// of Engine/RoughTransmittance.h:11-19 (checked against sizeof), and calls the two const member functions on it.  This file contains no reference source.
#include <Engine/RoughTransmittance.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>

using namespace CudaTracerLib;

namespace {
struct layout {   // Engine/RoughTransmittance.h:11-19
    size_t etaSamples, alphaSamples, thetaSamples; float etaMin, etaMax, alphaMin, alphaMax; size_t transSize, diffTransSize; float *transDevice, *diffTransDevice, *transHost, *diffTransHost;
};
static_assert(sizeof(layout) == sizeof(RoughTransmittance), "member layout of RoughTransmittance");
struct table_box {
    RoughTransmittance* rt;
    table_box(const float* trans, const float* diff, uint32_t n_eta, uint32_t n_alpha, uint32_t n_theta, float eta_min, float eta_max, float alpha_min, float alpha_max) {
        layout L; std::memset(&L, 0, sizeof L);
        L.etaSamples = n_eta; L.alphaSamples = n_alpha; L.thetaSamples = n_theta; L.etaMin = eta_min; L.etaMax = eta_max; L.alphaMin = alpha_min; L.alphaMax = alpha_max;
        L.transSize = (size_t)2 * n_eta * n_alpha * n_theta; L.diffTransSize = (size_t)2 * n_eta * n_alpha;
        L.transHost = const_cast<float*>(trans); L.diffTransHost = const_cast<float*>(diff);     // the host build reads m_transHost / m_diffTransHost (RoughTransmittance.cu:64-68, 101-105)
        rt = (RoughTransmittance*)std::malloc(sizeof(RoughTransmittance)); std::memcpy((void*)rt, &L, sizeof L);
    }
    ~table_box() { std::free(rt); }
};
}  // namespace

namespace CudaTracerLib { RoughTransmittance* ref_rough_manager_objects(); }   // the accessor `make ref` appends to the extract: RoughTransmittanceManager's file-static m_sObjectsHost[3]

extern "C" {

// RoughTransmittanceManager::StaticInitialize's effect without its file and CUDA plumbing: slot 0 / 1 / 2 = beckmann.dat / phong.dat / ggx.dat (RoughTransmittance.cu:126-128), looked up
// by the distribution TYPE (Beckmann 0, GGX 1, Phong 2: :140-158) — the reference's own mismatch, reproduced by whoever fills the slots.  The arrays must stay alive.
int ref_rough_manager_set(int slot, const float* trans, const float* diff, uint32_t n_eta, uint32_t n_alpha, uint32_t n_theta, float eta_min, float eta_max, float alpha_min, float alpha_max) {
    if (slot < 0 || slot > 2 || n_eta < 2 || n_alpha < 2 || n_theta < 2) return -1;
    table_box B(trans, diff, n_eta, n_alpha, n_theta, eta_min, eta_max, alpha_min, alpha_max);
    std::memcpy((void*)&ref_rough_manager_objects()[slot], (const void*)B.rt, sizeof(RoughTransmittance));
    return 0;
}
float ref_rough_manager_eval(int type, float cosTheta, float alpha, float eta) { return RoughTransmittanceManager::Evaluate((MicrofacetDistribution::EType)type, cosTheta, alpha, eta); }
float ref_rough_manager_eval_diffuse(int type, float alpha, float eta) { return RoughTransmittanceManager::EvaluateDiffuse((MicrofacetDistribution::EType)type, alpha, eta); }

// trans: (2 * n_eta, n_alpha, n_theta) floats, diff: (2 * n_eta, n_alpha); queries: nq x {cosTheta, alpha, eta}; out: nq x {Evaluate, EvaluateDiffuse}
int ref_rough_transmittance_eval(const float* trans, const float* diff, uint32_t n_eta, uint32_t n_alpha, uint32_t n_theta, float eta_min, float eta_max, float alpha_min, float alpha_max,
                                 int nq, const float* queries, float* out) {
    if (n_eta < 2 || n_alpha < 2 || n_theta < 2) return -1;
    table_box B(trans, diff, n_eta, n_alpha, n_theta, eta_min, eta_max, alpha_min, alpha_max);
    for (int i = 0; i < nq; i++) {
        out[2 * i + 0] = B.rt->Evaluate(queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]);
        out[2 * i + 1] = B.rt->EvaluateDiffuse(queries[3 * i + 1], queries[3 * i + 2]);
    }
    return 0;
}

}  // extern "C"
