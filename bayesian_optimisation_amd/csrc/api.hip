// Small host-only entry points of libgpbo (version, error strings, padding rule).
#include "gpbo_internal.h"

extern "C" int gpbo_version(void) { return GPBO_VERSION; }

extern "C" const char *gpbo_strerror(int status) {
    switch (status) {
        case GPBO_OK: return "ok";
        case GPBO_ERR_ARG: return "invalid argument (null pointer, size, alignment or unsupported feature count)";
        case GPBO_ERR_LAUNCH: return "HIP launch/runtime error";
        case GPBO_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

extern "C" int64_t gpbo_padded_n(int64_t N) {
    if (N < 1) N = 1;
    return (N + GPBO_NPAD - 1) / GPBO_NPAD * GPBO_NPAD;
}

// ---- optional per-launch timing of the dominant kernel (hipEvents on the caller's stream) ----------
extern "C" int gpbo_profile_create(int32_t capacity, gpbo_profile **out) {
    if (capacity < 1 || !out) return GPBO_ERR_ARG;
    gpbo_profile *p = new gpbo_profile;
    p->capacity = capacity;
    p->count = 0;
    p->begin = new void *[capacity];
    p->end = new void *[capacity];
    p->cands = new int64_t[capacity];
    p->kbegin = new void *[capacity];
    p->kmode = new int32_t[capacity];
    p->qend = new void *[capacity];
    p->qmode = new int32_t[capacity];
    for (int i = 0; i < capacity; ++i) {
        hipEvent_t a, b, c, q;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess || hipEventCreate(&c) != hipSuccess ||
            hipEventCreate(&q) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        p->begin[i] = a;
        p->end[i] = b;
        p->kbegin[i] = c;
        p->kmode[i] = 0;
        p->qend[i] = q;
        p->qmode[i] = 0;
        p->cands[i] = 0;
    }
    *out = p;
    return GPBO_OK;
}

extern "C" void gpbo_profile_reset(gpbo_profile *p) {
    if (!p) return;
    for (int i = 0; i < p->count; ++i) p->qmode[i] = 0;   // (only gpbo_posterior_qei_f64 sets it)
    p->count = 0;
}

extern "C" int gpbo_profile_read(gpbo_profile *p, double *total_ms, int64_t *launches, int64_t *cands) {
    if (!p || !total_ms || !launches || !cands) return GPBO_ERR_ARG;
    double sum = 0.0;
    int64_t nc = 0;
    for (int i = 0; i < p->count; ++i) {
        hipEvent_t a = reinterpret_cast<hipEvent_t>(p->begin[i]), b = reinterpret_cast<hipEvent_t>(p->end[i]);
        if (hipEventSynchronize(b) != hipSuccess) return GPBO_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return GPBO_ERR_LAUNCH;
        sum += ms;
        nc += p->cands[i];
    }
    *total_ms = sum;
    *launches = p->count;
    *cands = nc;
    return GPBO_OK;
}

extern "C" int gpbo_profile_read_kstar(gpbo_profile *p, double *total_ms, int64_t *launches, int64_t *cands) {
    if (!p || !total_ms || !launches || !cands) return GPBO_ERR_ARG;
    double sum = 0.0;
    int64_t nc = 0, nl = 0;
    for (int i = 0; i < p->count; ++i) {
        if (p->kmode[i] == 0 || (p->kmode[i] == 2 && i == 0)) continue;
        hipEvent_t a = reinterpret_cast<hipEvent_t>(p->kmode[i] == 2 ? p->end[i - 1] : p->kbegin[i]);
        hipEvent_t b = reinterpret_cast<hipEvent_t>(p->begin[i]);
        if (hipEventSynchronize(b) != hipSuccess) return GPBO_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return GPBO_ERR_LAUNCH;
        sum += ms;
        nc += p->cands[i];
        ++nl;
    }
    *total_ms = sum;
    *launches = nl;
    *cands = nc;
    return GPBO_OK;
}

extern "C" int gpbo_profile_read_qei(gpbo_profile *p, double *total_ms, int64_t *launches, int64_t *cands) {
    if (!p || !total_ms || !launches || !cands) return GPBO_ERR_ARG;
    double sum = 0.0;
    int64_t nc = 0, nl = 0;
    for (int i = 0; i < p->count; ++i) {
        if (p->qmode[i] != 1) continue;
        hipEvent_t a = reinterpret_cast<hipEvent_t>(p->end[i]), b = reinterpret_cast<hipEvent_t>(p->qend[i]);
        if (hipEventSynchronize(b) != hipSuccess) return GPBO_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return GPBO_ERR_LAUNCH;
        sum += ms;
        nc += p->cands[i];
        ++nl;
    }
    *total_ms = sum;
    *launches = nl;
    *cands = nc;
    return GPBO_OK;
}

extern "C" void gpbo_profile_destroy(gpbo_profile *p) {
    if (!p) return;
    for (int i = 0; i < p->capacity; ++i) {
        (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(p->begin[i]));
        (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(p->end[i]));
        (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(p->kbegin[i]));
        (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(p->qend[i]));
    }
    delete[] p->begin;
    delete[] p->end;
    delete[] p->cands;
    delete[] p->kbegin;
    delete[] p->kmode;
    delete[] p->qend;
    delete[] p->qmode;
    delete p;
}
