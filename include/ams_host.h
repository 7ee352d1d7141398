/* libams_host.so -- host-side helpers of the hot path's callers, plain C, no GPU and no dependencies.

   The reference has no FFI; these replace pure-Python/numpy host work that sits on the step's critical path:

   ams_crc32c           TFRecord framing check of the input pipeline (data/dataset.py:435-438 writes, :444-530 reads records whose
                        length and payload carry masked CRC-32C; TF computes it in C++)
   ams_mt_choice_rows   the k-means restart seeds of models/Kmeans_2.py:61-66,
                            np.array([np.random.choice(range(l), size=C, replace=False) for _ in range(b)])
                        drawn from numpy's GLOBAL legacy MT19937 generator (seeded at models/network.py:17-18)
*/
#ifndef AMS_HOST_H
#define AMS_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* CRC-32C (Castagnoli, reflected, init/xorout 0xffffffff) of n bytes.  Check value: "123456789" -> 0xE3069283. */
uint32_t ams_crc32c(const uint8_t* p, size_t n);

/* R rows of np.random.choice(l, size=C, replace=False) == np.random.choice(range(l), ...) of numpy's legacy RandomState:
   permutation(l)[:C], i.e. Fisher-Yates from the top with rejection-sampled 32-bit MT19937 draws.
     key[624], *pos   numpy's MT19937 state (np.random.get_state()[1], [2]); updated IN PLACE to the state numpy would be in after
                      the R calls, so the caller can hand it back with np.random.set_state()
     out              int32 [R, C], row-major
   Bit-identical to numpy (tests/test_host_mirror.py::test_kmeans_reference_seeding*).  l <= 2^31.
   Returns 0; -1 on invalid arguments (C > l, pos outside [0, 624], null pointers); -2 if the scratch allocation fails. */
int ams_mt_choice_rows(uint32_t* key, int32_t* pos, int R, int l, int C, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
