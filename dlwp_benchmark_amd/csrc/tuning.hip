// Every override of a dispatch heuristic in one documented registry (round 4; the round-3 library read 28 environment variables
// at 31 places).  A knob is identified by its name; its value comes from dlwp_set_tuning(name, value) or, when that was never
// called for the name, from the environment variable DLWP_<NAME> (looked up at the time of use: measurement scripts and tests
// may change it between calls; an integer is taken as it is, any other non-empty string as 1).  Unset knobs mean "the library's
// own choice".  None of them changes results beyond summation order; they exist for A/B measurements and for tests that force a
// kernel variant.  dlwp_tuning_list enumerates names and meanings.
#include <climits>
#include <cstdlib>
#include <cstring>
#include "common.hip.h"
#include "dlwpmi_internal.h"

namespace {

struct Knob {
    const char* name;
    const char* doc;
    int value;
    bool set;
};

Knob g_knobs[] = {
    {"GEMM_NOGLDS", "1: never use the 128 x 128 LDS-DMA bf16 GEMM kernels (activation products and weight gradients)", 0, false},
    {"GEMM_NOGLDS_TN", "1: never use the LDS-DMA weight-gradient kernel", 0, false},
    {"GEMM_GLDS_FORCE", "1: use the LDS-DMA kernel wherever its operand layout allows (skip the shape heuristic)", 0, false},
    {"GEMM_GLDS_MINK", "least K for the 128 x 128 LDS-DMA activation kernels to be taken by shape (default 96)", 0, false},
    {"GEMM_GLDS_MINTILES", "least number of 128 x 128 output tiles for those kernels to be taken by shape (default 256)", 0, false},
    {"GEMM_GLDS_N96", "128 x 96 output tiles in those kernels: 0 never, 1 when N % 96 == 0 and whole rounds x tile width is lower (default), 2 wherever N % 96 == 0", 0, false},
    {"GEMM_GLDS_N96_TIE_K", "128 x 96 tiles also where rounds x width ties with 128-wide tiles, for products with K up to this (default 192)", 0, false},
    {"GEMM_GLDS_KD", "K-step depth of the LDS-DMA activation kernels: 32 or 64 (default by shape)", 0, false},
    {"GEMM_GLDS_TN_KD", "K-step depth of the LDS-DMA weight-gradient kernel: 32 or 64 (default by tile count)", 0, false},
    {"GEMM_GLDS_TN_WGS", "resident workgroup slots the weight-gradient kernel sizes its K slices for", 0, false},
    {"GEMM_TN_ATOMIC", "1: weight-gradient slices combined by float atomics instead of the slab + ordered reduction", 0, false},
    {"GEMM_TN_SLAB_MIN", "least number of output tiles for the slab reduction (default 24)", 0, false},
    {"GEMM_P8", "1: use the 256 x 256 two-group kernel wherever it applies", 0, false},
    {"GEMM_P8_STAGED", "1: the 256 x 256 kernel's four-pass LDS-staged epilogue instead of the register epilogue", 0, false},
    {"GEMM_P8_MINK", "least K for the 256 x 256 kernel (default 2048)", 0, false},
    {"GEMM_P8_FILL", "least fill (percent) of the 256 x 256 kernel's last round of workgroups for the kernel to be taken by shape (default 70)", 0, false},
    {"GEMM_P8_WIDE", "0: the 256 x 256 kernel's register epilogue with 8-byte / 16-byte pieces per lane instead of 128-byte rows (round-5 form)", 0, false},
    {"GEMM_NOP8_TN", "1: never use the 256 x 256 weight-gradient kernel", 0, false},
    {"GEMM_TILE", "64 or 128: tile edge of the register-staged GEMM (default by shape)", 0, false},
    {"GEMM_TILE128_MINN", "narrowest output the 128 x 128 GEMM tile is considered for (default 128)", 0, false},
    {"GEMM_XCD_SPLITK", "bit 0 (default on): split-K products of the register-staged GEMM get slice counts that are multiples of 8 and a slice's tiles on one XCD; bit 1 (off: measured -1.7 % on Pangu C4): the same for the LDS-DMA weight-gradient kernel", 0, false},
    {"GEMM_TRACE", "1: print one line per GEMM call (shape census of an eager step)", 0, false},
    {"GEMM_NOGROUP", "1: products parked by dlwp_gemm_group_begin / dlwp_weight_grad_group launch one by one", 0, false},
    {"WGRAD_GROUP_MAXT", "most tokens for the grouped weight-gradient launch (default 65536)", 0, false},
    {"WGRAD_GROUP_WGS", "workgroups the grouped weight-gradient launch shares among its products (default 896)", 0, false},
    {"WGRAD_MULTI_WGS", "resident workgroup slots dlwp_wgrad_segments sizes its K slices for (default 384)", 0, false},
    {"LN_BWD_WANT", "workgroups of the scalar LayerNorm backward kernel", 0, false},
    {"LN_BWD_NOWIDE", "1: never use the wide-row LayerNorm backward kernel", 0, false},
    {"LN_BWD_NW", "waves per workgroup of the narrow-row LayerNorm backward kernel: 4 or 8 (default: 8 for inputs of at least 2048 rows and 2 M elements, one workgroup per CU)", 0, false},
    {"LN_FWD_V3", "widths 96 / 192 / 384 / 768 on the three-chunk LayerNorm forward kernel: 0 never, 1 from 4 M elements (default), 2 from 2048 rows", 0, false},
    {"LN_BWD_V3", "0: widths 96 / 192 / 384 stay on the one-chunk / wide-row LayerNorm backward kernels instead of the three-chunk kernel; 2: width 768 takes it too", 0, false},
    {"LN_BWD_WGS", "workgroups of the wide-row LayerNorm backward kernel (default 384)", 0, false},
    {"WINATTN_D48", "0: head dims 33 .. 48 stay on the tiled window-attention kernels (default 1: windows of at most 64 tokens take the wave-per-window forward and the LDS-staged two-pass backward in the bf16 matrix mode)", 0, false},
    {"WINATTN_SMALL_MIN_PAIRS", "least (window, head) pairs for the head-dim-48 route of the wave-per-window family (default 1024)", 0, false},
    {"WINATTN_TILED", "1: the tiled window-attention kernels also for many short windows", 0, false},
    {"WINATTN_WG_BWD", "workgroup width selector of the wave-per-window attention backward", 0, false},
    {"WINATTN_NOLDS", "1: the register-fragment attention backward instead of the LDS-staged one", 0, false},
    {"WINATTN_DBG", "measurement only (wrong results): bit switches that drop phases of the one-pass attention backward", 0, false},
    {"WINATTN_FWD_LDS", "0: the wave-per-window token-layout attention forward instead of the LDS-staged one", 0, false},
    {"WINATTN_WG_FWD", "workgroup count of the LDS-staged attention forward", 0, false},
    {"WINATTN_BWD1P_SMALL", "1: the one-pass attention backward also for windows of at most 64 tokens", 0, false},
    {"WINATTN_BWD2PASS", "1: the two-pass LDS-staged attention backward instead of the one-pass kernel", 0, false},
    {"FFT_IBW", "inner lanes of the W-axis FFT pass", 0, false},
    {"FFT_IBH", "inner lanes of the H-axis FFT pass", 0, false},
    {"FFT_STATIC", "0: the run-time FFT plan also for the axis lengths that have a compile-time plan (W = 180, H = 90)", 0, false},
    {"FFT_SPW", "compile-time plan of a W axis of 180: 3 (15 x 12, 16 lanes, default), 1 (12 x 15), 5 (12 x 15, 32 lanes, 512 threads)", 0, false},
    {"FFT_SPH", "compile-time plan of an H axis of 90: 4 (10 x 9, 32 lanes, 320 threads, default), 2 (9 x 10), 6 (9 x 10, 16 lanes, 192 threads)", 0, false},
    {"FFT_PRIME_SYM", "0: an axis that is one odd prime (103) on the full-matrix pass instead of the folded cos / sin form", 0, false},
    {"FFT_NT512_FROM", "FFT tiles (signal length x inner lanes) above this size run with 512-thread workgroups (default 4096)", 0, false},
    {"FFT_SKIP_PASSES", "measurement only (wrong results): the FFT kernels skip their butterfly passes and only move data", 0, false},
    {"DHCONV_PACK", "1: the round-4 spectral-weight pack kernel (one read of the weight per image) instead of the block kernel", 0, false},
    {"DHCONV_APPLY", "1: the round-4 spectral-convolution kernel (one 256-row chunk per workgroup) instead of the pipelined one", 0, false},
    {"DHCONV_RC", "rows per chunk of the pipelined spectral-convolution kernel: 64 or 128 (default 128)", 0, false},
    {"CHAIN_ROT", "0: no rotation of the wave -> feature-tile assignment in the one-launch MLP chains (default 1)", 0, false},
};
constexpr int NKNOBS = sizeof(g_knobs) / sizeof(g_knobs[0]);

Knob* find(const char* name) {
    if (!name) return nullptr;
    if (!strncmp(name, "DLWP_", 5)) name += 5;
    for (int i = 0; i < NKNOBS; ++i)
        if (!strcmp(g_knobs[i].name, name)) return &g_knobs[i];
    return nullptr;
}

}  // namespace

int dlwp_tune(const char* name) {
    Knob* k = find(name);
    if (!k) return DLWP_TUNE_UNSET;                       // (an unregistered name is a programming error: dlwp_set_tuning refuses it)
    if (k->set) return k->value;
    char env[96] = "DLWP_";
    strncat(env, k->name, sizeof(env) - 6);
    const char* e = getenv(env);
    if (!e || !*e) return DLWP_TUNE_UNSET;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    return (end && *end == '\0') ? (int)v : 1;
}

extern "C" int dlwp_set_tuning(const char* name, int value) {
    Knob* k = find(name);
    DLWP_REQUIRE(k, DLWP_E_INVALID, "set_tuning: unknown knob '%s' (dlwp_tuning_list)", name ? name : "(null)");
    k->value = value;
    k->set = true;
    return DLWP_OK;
}

extern "C" int dlwp_clear_tuning(const char* name) {
    if (!name) {
        for (int i = 0; i < NKNOBS; ++i) g_knobs[i].set = false;
        return DLWP_OK;
    }
    Knob* k = find(name);
    DLWP_REQUIRE(k, DLWP_E_INVALID, "clear_tuning: unknown knob '%s'", name);
    k->set = false;
    return DLWP_OK;
}

extern "C" int dlwp_get_tuning(const char* name, int* value) {
    DLWP_REQUIRE(find(name) && value, DLWP_E_INVALID, "get_tuning: unknown knob '%s' / NULL result", name ? name : "(null)");
    const int v = dlwp_tune(name);
    *value = v == DLWP_TUNE_UNSET ? 0 : v;
    return v == DLWP_TUNE_UNSET ? 0 : 1;                  // 1: an override is in force, 0: the library's own choice
}

extern "C" int dlwp_tuning_list(int index, const char** name, const char** doc) {
    if (index < 0 || index >= NKNOBS) return 0;
    if (name) *name = g_knobs[index].name;
    if (doc) *doc = g_knobs[index].doc;
    return 1;
}
