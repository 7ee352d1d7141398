/* ams_bss.h -- C ABI of libams_bss.so: BSS-eval SDR / SIR / SAR on MI355X.
 *
 * Replaces the reference's cupy path  utils/bss_eval.py:586-748  (bss_eval_sources_cupy and helpers), the one
 * experiments/evaluation/eval.py:48-73 calls once per utterance for (non_mix, mix) and (non_mix, separated).
 * Kept in its own shared library because it links hipFFT and hipSOLVER (plain library FFT / Cholesky); the training and
 * inference library libams_hip.so has no such dependency.
 *
 * All arrays are float64 on the device (the reference casts to float64, :595-596).  One context = one geometry
 * (nsrc, nsampl, flen); it owns the hipFFT plans and the hipSOLVER handle, nothing else -- scratch comes from the caller.
 */
#ifndef AMS_BSS_H
#define AMS_BSS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ams_bss_ctx ams_bss_ctx;

/* 0 on success, <0 on error (-1 invalid argument, -2 workspace too small, -3 launch / library failure). */
int ams_bss_abi_version(void);
int ams_bss_create(ams_bss_ctx** out, int nsrc, int nsampl, int flen);
void ams_bss_destroy(ams_bss_ctx* ctx);
size_t ams_bss_workspace_bytes(const ams_bss_ctx* ctx);

/* ref, est: [nsrc, nsampl].  crit: [3, nsrc, nsrc] = sdr | sir | sar of every (estimate jest, reference jtrue) pair,
 * crit[k][jest][jtrue] (utils/bss_eval.py:603-611).  info[0] != 0 when a Gram matrix was not positive definite (a silent
 * reference): the criteria are then NaN, which eval.py:61-62 skips.  The permutation by best mean SIR (:613-620) is a
 * 2..6-element host loop and is left to the caller. */
int ams_bss_eval_pairs(ams_bss_ctx* ctx, const double* ref, const double* est, double* crit, int* info, void* ws, size_t ws_bytes,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif
