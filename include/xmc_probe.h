/* Diagnostic probes of the MI355X build -- NOT part of the product ABI (include/xmcgan_hip.h, libxmcgan_hip.so).
 * Built into a separate libxmc_probe.so (csrc_probe/); used by tests/test_gpu_kernels.py::test_probe_layouts, by bench.py's
 * instrumented step (xmc_delay) and by the measurement scripts under tools/.  Same conventions as the product ABI: plain
 * pointers, explicit stream, 0 / negative errno-style return codes, no allocation. */
#ifndef XMC_PROBE_H_
#define XMC_PROBE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Dumps MFMA fragment / ds_read_b64_tr_b16 lane maps (tests/test_gpu_kernels.py). out: 2*64*16 + 64*4
 * floats. */
int xmc_probe_layouts(float* out, void* stream);
/* Registers-only MFMA loop (mode bit 0: 0 = v_mfma_f32_32x32x16_bf16, 1 = v_mfma_scale_f32_32x32x64_f8f6f4; bit 1:
 * constant instead of pseudo-random operands -- the chip holds a higher clock when the multipliers do not toggle): `blocks`
 * workgroups of 4 waves, each wave issues 8 * iters MFMAs on 8 independent accumulators.  The sustained matrix-core
 * rate of the box at the clock it holds under that load (tools/mfma_rate_probe.py); out: >= 1 float, not written. */
int xmc_mfma_rate_probe(int32_t mode, int32_t blocks, int32_t iters, float* out, void* stream);
/* Keeps `stream` busy for the given time (one sleeping wave, <= 2 s): work enqueued behind it starts when it ends.  bench.py
 * puts it in front of its instrumented step so that the HOST runs ahead of the GPU -- with an empty queue a start event
 * executes the moment it is enqueued and the host's launch latency lands inside the measured interval (seen on GPU boxes
 * with slow hosts: the 3x3 launches "took" 18.8 instead of 15.3 ms per step). */
int xmc_delay(int32_t microseconds, void* stream);
/* probe: `iters` rounds per lane of v_pk_mul_f32 + v_pk_add_f32 with crossed halves (mode 0) / uncrossed (mode 1), or of the broadcast forms of v_pk_fma_f32 the
 * epilogues use (mode 2: op_sel:[1,0,0], mode 3: op_sel_hi:[0,1,1] with an SGPR pair), against
 * scalar arithmetic; *bad (uint32, zeroed by the caller) += rounds whose bits differ. */
int xmc_pk_add_cross_probe(int32_t mode, int32_t blocks, int32_t iters, uint32_t* bad, void* stream);
/* probe neighbour: a busy loop of the instruction classes selected by `mask` (csrc/probe.hip lists the bits), to run on a second stream
 * beside xmc_pk_add_cross_probe; src: >= 1 MiB of device memory; out: >= 1 float, not written. */
int xmc_class_neighbour(int32_t mask, int32_t blocks, int32_t iters, const void* src, int64_t src_bytes, float* out, void* stream);
/* L2 -> CU delivery rate of the two load paths of the convolution kernels on a small, L2-resident region every
 * workgroup re-reads (tools/load_path_probe.py).  mode bit 0: 0 = global_load_dwordx4 into registers, 1 =
 * buffer_load_dwordx4 ... lds (LDS-DMA ring, counted vmcnt); bit 1: 0 = every instruction reads 1 KiB contiguous, 1 = 16
 * rows x 64 bytes at a 2 KiB stride; bits 4-7: depth in stages of 6 instructions per wave (2, 3 or 5).  Each of the
 * `blocks` workgroups (4 waves) moves iters * 24 KiB.  src_bytes in [1 MiB, 4 GiB); out: >= 1 float, not written. */
int xmc_load_path_probe(int32_t mode, int32_t blocks, int32_t iters, const void* src, int64_t src_bytes, float* out,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XMC_PROBE_H_ */
