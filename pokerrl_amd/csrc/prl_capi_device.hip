// C-ABI entry points that need the device: availability probe, batched hand evaluation (host and device pointer forms).
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "prl_cards.h"
#include "prl_device.h"
#include "prl_host.h"
#include "prl_kernels.h"
#include "prl_rt.h"

static std::mutex g_lut_mutex;
static uint16_t* g_d_hole_lut = nullptr;  // [1326] c1 | c2 << 8, device resident for the life of the process

int prl_hole_lut_device(const uint16_t** out) {
    std::lock_guard<std::mutex> lock(g_lut_mutex);
    if (!g_d_hole_lut) {
        std::vector<uint16_t> h(1326);
        int idx = 0;
        for (int c1 = 0; c1 < 52; ++c1)
            for (int c2 = c1 + 1; c2 < 52; ++c2) h[idx++] = (uint16_t)(c1 | (c2 << 8));
        PRL_HIP_TRY(hipMalloc((void**)&g_d_hole_lut, 1326 * sizeof(uint16_t)));
        PRL_HIP_TRY(hipMemcpy(g_d_hole_lut, h.data(), 1326 * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    *out = g_d_hole_lut;
    return PRL_OK;
}

extern "C" {

int32_t prl_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

int32_t prl_set_device(int32_t ordinal) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || ordinal < 0 || ordinal >= n) { prl_set_error("no such HIP device"); return PRL_ERR_NO_DEVICE; }
    PRL_HIP_TRY(hipSetDevice(ordinal));
    return PRL_OK;
}

const char* prl_build_flavor(void) {
#if defined(PRL_EMU)
    return "emu-host (tests only)";
#else
    return "hip-gfx950";
#endif
}

int32_t prl_hand_rank_boards_device(const void* d_boards_1d, int32_t n_boards, void* d_out_ranks, void* stream) {
    if (n_boards < 0 || (n_boards > 0 && (!d_boards_1d || !d_out_ranks))) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device"); return PRL_ERR_NO_DEVICE; }
    const uint16_t* lut = nullptr;
    int e = prl_hole_lut_device(&lut);
    if (e) return e;
    prl_launch_hand_rank_boards((const int8_t*)d_boards_1d, n_boards, lut, (int32_t*)d_out_ranks, stream);
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int32_t prl_hand_rank_boards(const int8_t* boards_1d, int32_t n_boards, int32_t* out_ranks) {
    if (n_boards < 0 || (n_boards > 0 && (!boards_1d || !out_ranks))) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device"); return PRL_ERR_NO_DEVICE; }
    if (n_boards == 0) return PRL_OK;
    for (int64_t i = 0; i < (int64_t)n_boards * 5; ++i)
        if (boards_1d[i] < 0 || boards_1d[i] >= 52) { prl_set_error("board card out of range"); return PRL_ERR_ARG; }
    int8_t* d_b = nullptr;
    int32_t* d_o = nullptr;
    // chunked so that a 2.6 M-board sweep needs a bounded staging buffer
    const int64_t chunk = 1 << 16;
    int64_t cap = n_boards < chunk ? n_boards : chunk;
    PRL_HIP_TRY(hipMalloc((void**)&d_b, (size_t)cap * 5));
    hipError_t he = hipMalloc((void**)&d_o, (size_t)cap * 1326 * sizeof(int32_t));
    if (he != hipSuccess) { (void)hipFree(d_b); prl_set_error(std::string("hipMalloc: ") + hipGetErrorString(he)); return PRL_ERR_OOM; }
    int rc = PRL_OK;
    for (int64_t s = 0; s < n_boards && rc == PRL_OK; s += cap) {
        int64_t n = n_boards - s < cap ? n_boards - s : cap;
        if (hipMemcpy(d_b, boards_1d + s * 5, (size_t)n * 5, hipMemcpyHostToDevice) != hipSuccess) { rc = PRL_ERR_HIP; break; }
        rc = prl_hand_rank_boards_device(d_b, (int32_t)n, d_o, nullptr);
        if (rc) break;
        if (hipMemcpy(out_ranks + s * 1326, d_o, (size_t)n * 1326 * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = PRL_ERR_HIP;
    }
    (void)hipFree(d_b);
    (void)hipFree(d_o);
    if (rc == PRL_ERR_HIP) prl_set_error("hipMemcpy failed");
    return rc;
}

int32_t prl_hand_rank_checksums(const int8_t* boards_1d, int32_t n_boards, int32_t chunk, uint64_t* out_checksums) {
    if (n_boards <= 0 || chunk <= 0 || !boards_1d || !out_checksums) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device"); return PRL_ERR_NO_DEVICE; }
    const uint16_t* lut = nullptr;
    int e = prl_hole_lut_device(&lut);
    if (e) return e;
    const int n_chunks = (n_boards + chunk - 1) / chunk;
    int8_t* d_b = nullptr;
    unsigned long long* d_o = nullptr;
    PRL_HIP_TRY(hipMalloc((void**)&d_b, (size_t)n_boards * 5));
    if (hipMalloc((void**)&d_o, (size_t)n_chunks * 8) != hipSuccess) { (void)hipFree(d_b); prl_set_error("hipMalloc failed"); return PRL_ERR_OOM; }
    int rc = PRL_OK;
    if (hipMemcpy(d_b, boards_1d, (size_t)n_boards * 5, hipMemcpyHostToDevice) != hipSuccess) rc = PRL_ERR_HIP;
    if (!rc) {
        prl_launch_hand_rank_checksums(d_b, n_boards, chunk, lut, d_o, nullptr);
        if (hipMemcpy(out_checksums, d_o, (size_t)n_chunks * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = PRL_ERR_HIP;
    }
    (void)hipFree(d_b);
    (void)hipFree(d_o);
    if (rc) prl_set_error("hip error in prl_hand_rank_checksums");
    return rc;
}

// CppHandeval.py:45-65 legacy form: row-pointer arrays; `out` rows are written in full (blocked hands = -1)
void get_hand_rank_all_hands_on_given_boards_52_holdem(int32_t** out, int8_t** boards_1d, int32_t n_boards,
                                                       int8_t** /*lut_idx_2_hole_cards*/, int8_t** /*lut_1d_2_2d*/) {
    if (n_boards <= 0) return;
    std::vector<int8_t> b((size_t)n_boards * 5);
    for (int i = 0; i < n_boards; ++i)
        for (int k = 0; k < 5; ++k) b[(size_t)i * 5 + k] = boards_1d[i][k];
    std::vector<int32_t> r((size_t)n_boards * 1326);
    if (prl_hand_rank_boards(b.data(), n_boards, r.data()) != PRL_OK) return;  // legacy signature has no error channel
    for (int i = 0; i < n_boards; ++i) memcpy(out[i], r.data() + (size_t)i * 1326, 1326 * sizeof(int32_t));
}

}  // extern "C"
