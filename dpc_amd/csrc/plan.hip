// plan.hip -- kernel-selection trace and query (include/dpc_hip.h: dpc_conv_plan, dpc_last_kernel).
//
// dpc_conv_igemm / dpc_conv_wgrad choose among a dozen kernels by shape, element type and size thresholds
// (conv_halo.hip, conv_igemm_ws.hip, conv_igemm.hip, conv_wgrad_patch.hip, conv_wgrad_stem.hip, conv_wgrad.hip).  A silent
// demotion of a shape to the generic kernel keeps every numerical test green and costs 2-3x at run time, so the choice is
// observable: every launch site records the kernel it selected (DPC_LAUNCH, dpc_rt.h), dpc_last_kernel() returns the record
// of the calling thread's most recent launch, and dpc_conv_plan() runs the SAME dispatch code in plan-only mode (dummy,
// 16-byte aligned pointers; the launch is skipped) -- the tests assert the variant each case is meant to cover.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

thread_local int dpc_tls_plan_only = 0;

// process-wide (the engine sets it around the part of the backward pass that overlaps the gradient all-reduce); kernels read it at
// plan time on the host, so a captured hipGraph keeps the grid it was captured with
static int g_reserved_cus = 0;
int dpc_reserved_cus() { return g_reserved_cus; }
extern "C" int dpc_set_reserved_cus(int32_t n) {
    if (n < 0 || n > 128) return DPC_ERR_ARG;
    const int prev = g_reserved_cus;
    g_reserved_cus = n;
    return prev;
}
// how the f32 kernels multiply (dpc_rt.h "bf16x6"): read at launch time on the host, travels as a kernel argument
static int g_f32_matmul = 0;
int dpc_f32_matmul_mode() { return g_f32_matmul; }
extern "C" int dpc_set_f32_matmul(int32_t mode) {
    if (mode != 0 && mode != 1) return DPC_ERR_ARG;
    const int prev = g_f32_matmul;
    g_f32_matmul = mode;
    return prev;
}
static thread_local char tls_name[192] = "";
static thread_local char tls_detail[96] = "";

// "(igemm_ws_kernel<true>)" -> "igemm_ws_kernel<true>": outer parentheses and blanks dropped; a detail recorded by the
// dispatcher BEFORE the launch site (template arguments that are only names at the launch site) is appended in brackets
void dpc_plan_note(const char* expr) {
    size_t n = 0;
    int depth = 0;
    for (const char* c = expr; *c && n + 1 < sizeof(tls_name); ++c) {
        if (*c == ' ') continue;
        if (*c == '(' && depth++ == 0) continue;
        if (*c == ')' && --depth == 0) continue;
        tls_name[n++] = *c;
    }
    tls_name[n] = 0;
    if (tls_detail[0] && n + strlen(tls_detail) + 3 < sizeof(tls_name)) {
        snprintf(tls_name + n, sizeof(tls_name) - n, "[%s]", tls_detail);
        tls_detail[0] = 0;
    }
}

void dpc_plan_detail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls_detail, sizeof(tls_detail), fmt, ap);
    va_end(ap);
}

static int copy_name(char* name, int32_t cap) {
    if (!name || cap <= 0) return DPC_ERR_ARG;
    const int n = (int)strlen(tls_name);
    snprintf(name, (size_t)cap, "%s", tls_name);
    return n;
}

extern "C" int dpc_last_kernel(char* name, int32_t cap) { return copy_name(name, cap); }

extern "C" int dpc_conv_plan(const dpc_conv_desc* d, int32_t op, int32_t flags, int32_t dy_ld, char* name, int32_t cap) {
    if (!d || !name || cap <= 0 || (op != DPC_PLAN_IGEMM && op != DPC_PLAN_WGRAD)) return DPC_ERR_ARG;
    void* const dummy = (void*)(uintptr_t)4096;  // never dereferenced: the launch is skipped
    tls_name[0] = 0;
    tls_detail[0] = 0;
    dpc_tls_plan_only = 1;
    int rc;
    if (op == DPC_PLAN_IGEMM && (flags & (DPC_PLAN_ADDEND_MASK | DPC_PLAN_BNRED))) {
        dpc_conv_epilogue e = {};
        const bool red = flags & DPC_PLAN_BNRED;
        e.addend = (flags & (DPC_PLAN_ADDEND | DPC_PLAN_ADDEND_MASK)) ? dummy : nullptr;
        e.addend_mask = (flags & DPC_PLAN_ADDEND_MASK) ? (const uint8_t*)dummy : nullptr;
        e.bn_raw = red ? dummy : nullptr;
        e.bn_mask = red ? (const uint8_t*)dummy : nullptr;
        e.bn_mean = e.bn_invstd = red ? (const float*)dummy : nullptr;
        e.stats = (red || (flags & DPC_PLAN_STATS)) ? (float*)dummy : nullptr;
        rc = dpc_conv_igemm_ex(d, dummy, dummy, dummy, &e, nullptr);
    } else if (op == DPC_PLAN_IGEMM) {
        rc = dpc_conv_igemm(d, dummy, dummy, dummy, (flags & DPC_PLAN_ADDEND) ? dummy : nullptr,
                            (flags & DPC_PLAN_STATS) ? (float*)dummy : nullptr, nullptr);
    } else {
        int32_t ns = 0;
        rc = dpc_conv_wgrad(d, dummy, dummy, dy_ld, (float*)dummy, &ns, nullptr);
    }
    dpc_tls_plan_only = 0;
    if (rc < 0) return rc;
    return copy_name(name, cap);
}
