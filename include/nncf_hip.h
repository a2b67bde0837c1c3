/*
 * nncf_hip.h -- C ABI of libnncf_hip.so, the MI355X (gfx950) implementation of the NNCF
 * sampling + scoring hot path.
 *
 * The reference (chentingpc/NNCF) has exactly one FFI on this path: the Cython class
 * `MultinomialSampler` (sampler/sampler.pyx:19-38) over `nodesampler::NodeSampler`
 * (sampler/nodesampler.cpp:16-77).  Everything else on the path is reached through
 * Keras `train_on_batch` / `predict_on_batch` on the dict returned by
 * `get_model` (models/model_framework.py:18,195-206).  Each entry point below names the
 * reference interface it replaces.  INTEGRATION.md shows the binding a maintainer of
 * the reference would add (ctypes stubs in place of `from sampler import
 * MultinomialSampler` and of the Keras model objects).
 *
 * Conventions
 *  - plain C types only; device buffers are raw pointers owned by the caller (e.g. the
 *    storage of a torch tensor); the library never frees caller memory;
 *  - every function returns 0 on success or a negative nncf_status; nncf_last_error()
 *    gives a thread-local message;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); no entry point
 *    synchronises the device unless its comment says so;
 *  - one handle = one sequential stream of state; handles are not thread-safe, distinct
 *    handles are independent.
 */
#ifndef NNCF_HIP_H_
#define NNCF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NNCF_OK = 0,
  NNCF_ERR_INVALID = -1,  /* bad argument (maps to ValueError) */
  NNCF_ERR_NOMEM = -2,    /* host or device allocation failed (MemoryError) */
  NNCF_ERR_HIP = -3,      /* HIP runtime error, no usable device (RuntimeError) */
  NNCF_ERR_STATE = -4     /* call order / handle state error (RuntimeError) */
} nncf_status;

const char* nncf_last_error(void);
/* Library version string, e.g. "nncf_hip 0.1 (gfx950)". */
const char* nncf_version(void);
/* Number of visible HIP devices (0 when there is none); never throws. */
int nncf_device_count(void);

/* ------------------------------------------------------------------------------------
 * B1  Negative sampler  --  replaces nodesampler::NodeSampler / MultinomialSampler
 *     (sampler/nodesampler.cpp:16-77, sampler/sampler.pyx:19-38)
 *
 * The reference materialises `int neg_table[1e8]` where id i owns a share proportional
 * to dist[i]^power (set_table, nodesampler.cpp:29-49) and draws
 * neg_table[((s = s*25214903917+11) >> 16) % 1e8]  (Rand, :23-26; sample_batch, :73-76).
 * The table is monotone in the id, so it is fully described by the int32 boundary array
 * bound[i] = #entries holding an id <= i.  The library builds bound[] on the host with
 * the reference's own double-precision predicate (bit-identical boundaries) in O(V), and
 * the device kernel maps each draw r to the id with bound[id-1] <= r < bound[id].
 * Sample streams are bit-identical to the reference for equal LCG state.
 * ---------------------------------------------------------------------------------- */
typedef struct nncf_sampler nncf_sampler;

/* NodeSampler::NodeSampler (nodesampler.cpp:55-67) + set_table (:29-49).
 * dist: host pointer, n doubles (degree / weight per id; zero => never sampled).
 * seed: LCG state before the first draw.  seed == 0 => time(NULL), as the reference
 *       (:59-60).  (The reference leaves the state uninitialised for seed != 0; here
 *       "state0 := seed" is the contract.)
 * device: HIP device ordinal that will own the boundary arrays.
 * Errors: n <= 0, n >= 1e8, NULL dist, negative/non-finite weight, all-zero dist
 *         -> NNCF_ERR_INVALID; no device -> NNCF_ERR_HIP. */
int nncf_sampler_create(const double* dist, int64_t n, double power, uint64_t seed, int device,
                        nncf_sampler** out);
int nncf_sampler_destroy(nncf_sampler* s);

/* Host-only half of nncf_sampler_create: the boundary array for `dist` (no device needed).
 * Restates set_table (nodesampler.cpp:29-49) in O(V); bit-identical to scanning the
 * reference's table. */
int nncf_sampler_build_bounds(const double* dist, int64_t n, double power, int32_t* out_host);

/* Copies bound[0..n) (int32, host) -- used to hash/compare against the reference table. */
int nncf_sampler_bounds(const nncf_sampler* s, int32_t* out_host);
int64_t nncf_sampler_size(const nncf_sampler* s);

/* NodeSampler::sample_batch (nodesampler.cpp:73-76), device output.
 * Writes n int32 ids to d_out (device memory) on `stream` and advances the LCG state by
 * n draws.  Asynchronous. */
int nncf_sampler_sample_dev(nncf_sampler* s, int64_t n, int32_t* d_out, void* stream);

/* Same draw, but only for positions [offset, offset+count) of the next n-draw batch;
 * does NOT advance the state.  Lets G ranks each produce their slice of one logical
 * stream (concatenation == the 1-GPU stream); callers then call nncf_sampler_skip(n). */
int nncf_sampler_sample_slice_dev(nncf_sampler* s, int64_t offset, int64_t count, int32_t* d_out,
                                  void* stream);
int nncf_sampler_skip(nncf_sampler* s, int64_t n);

/* MultinomialSampler.sample_batch (sampler.pyx:33-38): device draw + copy into a host
 * buffer of n int32; synchronises `stream`. */
int nncf_sampler_sample_batch(nncf_sampler* s, int64_t n, int32_t* out_host, void* stream);

/* LCG state access (the value `seed` holds in nodesampler.cpp:19). */
int nncf_sampler_get_state(const nncf_sampler* s, uint64_t* state);
int nncf_sampler_set_state(nncf_sampler* s, uint64_t state);

/* LCG jump-ahead helper: state after `n` draws from `state` (no sampler needed). */
uint64_t nncf_lcg_jump(uint64_t state, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* NNCF_HIP_H_ */
