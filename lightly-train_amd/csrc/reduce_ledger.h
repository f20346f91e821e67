// Deferred, order-fixed reductions ("ledger").  Kernels that end in a cross-workgroup sum (LayerNorm dgamma / dbeta, bias column sums,
// LayerScale gradients, the mask-token gradient) normally finish with one fp32 atomicAdd per column per workgroup: cheap, but the order
// of the adds -- and with it the last bits of the result -- changes from run to run.  Between lt_reduce_begin() and lt_reduce_end()
// they instead store their per-workgroup partial rows into a caller-owned scratch region and record (destination, partial rows); one
// lt_reduce_flush() launch then adds every destination's partial rows in record order with a fixed summation tree.  Bitwise
// reproducible, no same-address atomics, and no extra launch per producer.
#pragma once
#include <cstddef>

namespace lt_ledger {
bool active();
// `floats` of scratch (16-byte aligned), or nullptr when deferral is off or the scratch is exhausted (the caller falls back to atomics)
float* reserve(size_t floats);
// dst[c] += sum_p src[p * stride + c], c < C, p < nparts -- at the next flush
void record(float* dst, const float* src, int nparts, long stride, int C);
}  // namespace lt_ledger
