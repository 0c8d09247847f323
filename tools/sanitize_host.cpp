// Host-side sanitizer job: driver for a build of libgpbo's HOST code - every translation unit of csrc/
// compiled with `-fsanitize=address,undefined -fno-gpu-sanitize` (host code instrumented; tools/sanitize_host.sh) - on a machine WITHOUT a GPU.
// It exercises what runs on the host before any kernel is launched:
//   * the launch planner of the fused factorisation (cholinv_plan.h through gpbo_cholinv_plan) for every padded size
//     128 ... 16,384 with the option sets the tests and the product use, checking each plan's tiles for well-formedness;
//   * the argument validation of every compute entry point (the bad-argument table of tests/test_abi_cpu.py, plus
//     all-NULL calls), which must return an error code before touching a pointer or the HIP runtime.
// Sanitizers are for this CPU build only (no GPU AddressSanitizer on the pool).  Prints "sanitize_host ok: ..." and exits 0;
// a sanitizer finding aborts with its report (-fno-sanitize-recover).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/gpbo.h"

static int fails = 0;
#define EXPECT(cond)                                                  \
    do {                                                              \
        if (!(cond)) {                                                \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++fails;                                                  \
        }                                                             \
    } while (0)

static long plans = 0, tiles_seen = 0;

static void run_plan(int64_t Np, const int32_t *opt) {
    int64_t nl = 0, nt = 0;
    int rc = gpbo_cholinv_plan(Np, opt, &nl, &nt, nullptr, nullptr);
    EXPECT(rc == GPBO_OK);
    if (rc != GPBO_OK) return;
    EXPECT(nl >= 1 && nt >= 0);
    // exact-size arrays: an off-by-one write of the export loop lands in ASan's red zone
    std::vector<int32_t> L((size_t)(5 * nl)), T((size_t)(8 * nt));
    int64_t cl = nl, ct = nt;
    rc = gpbo_cholinv_plan(Np, opt, &cl, &ct, L.data(), T.empty() ? nullptr : T.data());
    EXPECT(rc == GPBO_OK && cl == nl && ct == nt);
    if (nt > 1) {  // one entry short: refused, nothing written past the end
        std::vector<int32_t> Ts((size_t)(8 * (nt - 1)));
        int64_t c2 = nl, c3 = nt - 1;
        EXPECT(gpbo_cholinv_plan(Np, opt, &c2, &c3, L.data(), Ts.data()) == GPBO_ERR_WORKSPACE);
    }
    int64_t covered = 0;
    int pairs = 0;
    for (int64_t i = 0; i < nl; ++i) {
        const int32_t *l = &L[(size_t)(5 * i)];
        EXPECT(l[2] >= 0 && l[3] >= 0 && (int64_t)l[2] + l[3] <= nt && l[4] >= 1);
        EXPECT(l[1] + l[3] > 0);
        if (l[1] > 0) {
            EXPECT(l[0] == pairs);
            ++pairs;
        }
        covered += l[3];
    }
    EXPECT(pairs == Np / 128);
    EXPECT(covered == nt);
    for (int64_t i = 0; i < nt; ++i) {
        const int32_t *t = &T[(size_t)(8 * i)];
        const int th = t[0] == 2 ? 64 : t[0] == 3 ? 128 : t[0] == 4 ? 256 : 0;
        const int tw = t[0] == 2 ? (t[7] == 32 ? 32 : 64) : 128;
        EXPECT(th != 0);
        EXPECT(t[1] >= 0 && t[2] >= 32 && t[2] % 32 == 0 && t[1] + t[2] <= t[3]);   // source rows are finished rows
        EXPECT(t[3] % 64 == 0 && t[3] < Np && t[5] > t[3] && t[5] <= Np);
        EXPECT(t[4] >= 0 && t[4] % tw == 0 && (int64_t)t[4] + tw <= 2 * Np);
        EXPECT(t[6] >= 0 && t[6] <= Np);
    }
    ++plans;
    tiles_seen += nt;
}

int main() {
    EXPECT(gpbo_version() == GPBO_VERSION);
    EXPECT(gpbo_strerror(GPBO_ERR_ARG) != nullptr && gpbo_strerror(-99) != nullptr && gpbo_strerror(0) != nullptr);
    for (int64_t n = -3; n < 70000; n += 97) {
        const int64_t p = gpbo_padded_n(n);
        if (n >= 1) EXPECT(p >= n && p % GPBO_NPAD == 0 && p - n < GPBO_NPAD);
    }

    // ---- planner: every size, default options; the option sets of tests/test_cholinv_plan_cpu.py at every size they fit
    static const int32_t OPTS[][7] = {
        {1, 128, 3, 1, 0, 0, 0}, {2, 256, 4, 2, 0, 0, 0}, {3, 384, 3, 2, 0, 0, 0}, {1, 256, 4, 1, 0, 0, 0},
        {2, 512, 4, 3, 0, 0, 0}, {3, 256, 4, 3, 0, 0, 0}, {1, 128, 3, 1, 0, 2, 0}, {1, 256, 3, 3, 0, 0, 64},
        {1, 384, 3, 3, 0, 0, 64}, {1, 256, 3, 3, 0, 0, 32}, {4, 768, 3, 1, 0, 5, 32},
    };
    for (int64_t Np = 128; Np <= 16384; Np += 128) {
        run_plan(Np, nullptr);
        if (Np <= 4096 || Np % 2048 == 0)
            for (const auto &o : OPTS) run_plan(Np, o);
    }
    {  // refused before any planning: sizes, a far rank that is no multiple of 128, a tile width that does not exist
        int64_t a = 0, b = 0;
        EXPECT(gpbo_cholinv_plan(100, nullptr, &a, &b, nullptr, nullptr) == GPBO_ERR_ARG);
        EXPECT(gpbo_cholinv_plan(0, nullptr, &a, &b, nullptr, nullptr) == GPBO_ERR_ARG);
        EXPECT(gpbo_cholinv_plan(-128, nullptr, &a, &b, nullptr, nullptr) == GPBO_ERR_ARG);
        EXPECT(gpbo_cholinv_plan((int64_t)1 << 40, nullptr, &a, &b, nullptr, nullptr) == GPBO_ERR_ARG);
        EXPECT(gpbo_cholinv_plan(32768 + 128, nullptr, &a, &b, nullptr, nullptr) == GPBO_ERR_ARG);
        EXPECT(gpbo_cholinv_plan(1024, nullptr, nullptr, &b, nullptr, nullptr) == GPBO_ERR_ARG);
        const int32_t bad1[7] = {0, 200, 0, 0, 0, 0, 0}, bad2[7] = {0, 0, 0, 0, 0, 0, 48}, bad3[7] = {65, 0, 0, 0, 0, 0, 0},
                      bad4[7] = {0, 0, 9, 0, 0, 0, 0}, bad5[7] = {0, 1 << 30, 0, 0, 0, 0, 0};
        for (const int32_t *o : {bad1, bad2, bad3, bad4, bad5}) EXPECT(gpbo_cholinv_plan(1024, o, &a, &b, nullptr, nullptr) == GPBO_ERR_ARG);
    }

    // ---- argument validation: the table of tests/test_abi_cpu.py::test_size_contracts_are_checked_on_the_host -------------
    alignas(256) static char buf[1024];
    void *p = buf;            // aligned like a device allocation; never dereferenced by a call that is refused
    double *pd = reinterpret_cast<double *>(buf);
    int32_t *pi = reinterpret_cast<int32_t *>(buf);
    double ls[16];
    for (double &v : ls) v = 0.5;
    EXPECT(gpbo_kxx_f64(nullptr, 4, 2, nullptr, 1e-4, 1e-6, nullptr, 128, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_posterior_workspace_bytes(100, 512, 10) == -1);
    EXPECT(gpbo_posterior_workspace_bytes(128, 500, 10) == -1);
    EXPECT(gpbo_posterior_workspace_bytes(128, 512, 1000) > 128 * 512 * 8);
    EXPECT(gpbo_factorise_workspace_bytes(256) == 8 * (2 * 256 * 256 + 256 * 64 + 256));
    EXPECT(gpbo_factorise_f64(pd, pd, 100, 2, ls, 1e-4, 1e-6, 256, pd, pd, pd, pi, p, (int64_t)1 << 40, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_factorise_f64(pd, pd, 100, 2, ls, 1e-4, 1e-6, 128, pd, pd, pd, pi, p, 8, nullptr) == GPBO_ERR_WORKSPACE);
    EXPECT(gpbo_append_f64(pd, pd, 128, 2, ls, 1e-4, 1e-6, 128, pd, pd, nullptr, pd, pd, pi, p, 1 << 30, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_append_f64(pd, pd, 10, 17, ls, 1e-4, 1e-6, 128, pd, pd, nullptr, pd, pd, pi, p, 1 << 30, nullptr) == GPBO_ERR_ARG);
    const double bad_ls[2] = {0.5, 0.0};
    EXPECT(gpbo_append_f64(pd, pd, 10, 2, bad_ls, 1e-4, 1e-6, 128, pd, pd, nullptr, pd, pd, pi, p, 1 << 30, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_append_f64(pd, pd, 10, 2, ls, 1e-4, 1e-6, 128, pd, pd, nullptr, pd, pd, pi, p, 8, nullptr) == GPBO_ERR_WORKSPACE);
    EXPECT(gpbo_append_workspace_bytes(128) == 8 * (3 * 128 + 8));
    auto post = [&](int64_t Np, int64_t chunk, int kind, int64_t wbytes) {
        return gpbo_posterior_acq_f64(pd, 1000, pd, 100, Np, 2, ls, pd, pd, 1.0, kind, 4.0, 0.0, 0.0, 0, chunk, nullptr, nullptr,
                                      nullptr, reinterpret_cast<gpbo_result *>(p), p, wbytes, nullptr, nullptr);
    };
    EXPECT(post(100, 512, 0, (int64_t)1 << 40) == GPBO_ERR_ARG);
    EXPECT(post(128, 500, 0, (int64_t)1 << 40) == GPBO_ERR_ARG);
    EXPECT(post(128, 512, 7, (int64_t)1 << 40) == GPBO_ERR_ARG);
    EXPECT(post(128, (int64_t)1 << 25, 0, (int64_t)1 << 40) == GPBO_ERR_ARG);
    EXPECT(post(128, 512, 0, 8) == GPBO_ERR_WORKSPACE);
    EXPECT(gpbo_potrf_f64(pd, 100, pd, pi, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_trtri_f64(pd, pd, 100, pd, pd, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_nlml_grid_f64(pd, pd, 177, 2, pd, 4, 1e-4, reinterpret_cast<float *>(p), nullptr) == GPBO_ERR_ARG);
    // the fused factorisation: S must be 16-byte aligned (LDS-DMA reads it in 16-byte pieces), ld even and >= 2 Np
    EXPECT(gpbo_cholinv_f64(pd + 1, 512, 256, pi, nullptr, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_cholinv_f64(pd, 511, 256, pi, nullptr, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_cholinv_f64(pd, 256, 256, pi, nullptr, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_cholinv_f64(pd, 512, 200, pi, nullptr, nullptr) == GPBO_ERR_ARG);
    EXPECT(gpbo_cholinv_f64(pd, 2 * 32896, 32896, pi, nullptr, nullptr) == GPBO_ERR_ARG);
    {
        const int32_t tile_bad[8] = {3, 0, 100, 128, 0, 256, 0, 0};  // rank not a multiple of 32
        EXPECT(gpbo_cholinv_tiles_f64(pd, 512, 256, pi, -1, tile_bad, 1, 1, 1, nullptr) == GPBO_ERR_ARG);
        EXPECT(gpbo_cholinv_tiles_f64(pd, 512, 256, pi, 5, nullptr, 0, 1, 1, nullptr) == GPBO_ERR_ARG);  // pair beyond Np
        EXPECT(gpbo_cholinv_tiles_f64(pd, 512, 256, pi, -1, nullptr, 0, 1, 1, nullptr) == GPBO_ERR_ARG);  // nothing to do
    }
    // host-pointer entry points: refused before the first HIP call
    gpbo_result res;
    int32_t info_h = 0;
    EXPECT(gpbo_select_next_host_f64(nullptr, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr,
                                     nullptr, &res, &info_h) == GPBO_ERR_ARG);
    EXPECT(gpbo_select_qei_host_f64(nullptr, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, &res,
                                    &info_h) == GPBO_ERR_ARG);
    EXPECT(gpbo_nlml_grid_host_f64(nullptr, nullptr, 0, 0, nullptr, 0, 0, nullptr) == GPBO_ERR_ARG);
    if (fails) {
        std::fprintf(stderr, "sanitize_host: %d check(s) failed\n", fails);
        return 1;
    }
    std::printf("sanitize_host ok: %ld plans (%ld tiles) built and checked, argument tables refused on the host\n", plans, tiles_seen);
    return 0;
}
