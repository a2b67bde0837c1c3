// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product path.
//
// Thin C-ABI shim that compiles the UNMODIFIED reference sampler where it lies
// (/root/reference/sampler/nodesampler.cpp, passed in as -DNNCF_REF_SRC=...) so that the
// CPU restatement in nodesampler_oracle.c and the HIP sampler can be pinned against the
// real thing.  No reference source text is copied here: the file is #include'd from its
// original location at build time and only the resulting .so lands in oracle/_ref/.
//
// The reference never assigns `seed` when rand_seed != 0 (nodesampler.cpp:59-60), so a
// reproducible stream needs the LCG state pinned from outside.  `class` members are
// private by default; redefining the keyword to `struct` AFTER the standard headers are in
// (their include guards then make the reference's own #includes no-ops) exposes
// `seed` and `neg_table` without touching the reference file.
#include <cstdlib>
#include <cstdio>
#include <cassert>
#include <cmath>
#include <ctime>
#include <new>

#define class struct
#include NNCF_REF_SRC
#undef class

extern "C" {

// state0 becomes the LCG state before the first draw ("seed := state0").
void* ref_sampler_create(const double* dist, int n, double power, unsigned long long state0) {
  // rand_seed=1 keeps the constructor away from time(NULL); state is pinned right after.
  nodesampler::NodeSampler* s = new (std::nothrow) nodesampler::NodeSampler(dist, n, power, 1ULL);
  if (!s) return nullptr;
  s->seed = state0;
  return s;
}

int ref_sampler_sample(void* h) { return static_cast<nodesampler::NodeSampler*>(h)->sample(); }

void ref_sampler_sample_batch(void* h, int n, int* out) {
  static_cast<nodesampler::NodeSampler*>(h)->sample_batch(n, out);
}

unsigned long long ref_sampler_get_state(void* h) {
  return static_cast<nodesampler::NodeSampler*>(h)->seed;
}

void ref_sampler_set_state(void* h, unsigned long long s) {
  static_cast<nodesampler::NodeSampler*>(h)->seed = s;
}

// Raw view of the 1e8-entry table (read-only use by the fixture generator).
const int* ref_sampler_table(void* h) { return static_cast<nodesampler::NodeSampler*>(h)->neg_table; }

long long ref_sampler_table_size(void) { return nodesampler::neg_table_size; }

// The reference class has no destructor (400 MB leak per instance); free the table here.
void ref_sampler_destroy(void* h) {
  nodesampler::NodeSampler* s = static_cast<nodesampler::NodeSampler*>(h);
  delete[] s->neg_table;
  delete s;
}

}  // extern "C"
