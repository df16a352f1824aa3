/* pais_test_hooks.h -- TEST AND MEASUREMENT HOOKS of libpais_hip.so.
 *
 * NOT part of the drop-in boundary: nothing here replaces a function of the reference, and a maintainer binding the
 * library includes pais_mvs.h / pais_hip.h only.  The two entry points exist so that
 *   - the host scheduler can be tested on machines without a GPU (records supplied by the test's checker), and
 *   - the sharded multi-GPU code path can be timed on ONE GPU (bench.py --emulate-world).
 * They are exported from the same shared object because they exercise its internals; neither computes a record on the host.
 */
#ifndef PAIS_TEST_HOOKS_H
#define PAIS_TEST_HOOKS_H

#include "pais_mvs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* MEASUREMENT AID: one rank of a larger world on ONE GPU (bench.py --emulate-world).  mode 1 on a single-rank driver records
 * the records of every batch, keyed by candidate; mode 2 then makes this driver rank `rank` of `world`: every sharded batch
 * runs the real sharded code path (shard refined on the GPU, packed, status header, copy down, unpack, replicated commit;
 * thin batches replicated; large rounds streamed) with the other ranks' blocks replayed from the recorded run in place of the
 * ncclAllGather -- the bytes arrive over PCIe instead of xGMI, plus PAIS_EMU_LATENCY_US (25) of modelled collective launch
 * latency.  mode 0 switches it off.  What it measures: T_rank(world) on this GPU; what it cannot: link contention and the
 * wait for the slowest rank (bench.py takes the maximum over the ranks it emulates). */
int  pais_mvs_emulate(pais_mvs *m, int mode, int rank, int world);

/* (GPU-less drivers only -- schedulers under test, a host that keeps the records elsewhere; a driver that owns a GPU context
 * refuses it.)  A driver created with device < 0 owns no GPU and never computes a record.  This callback lets the owner of the
 * records (a process that has the GPU, a test's checker) feed the monolithic entry points above -- the stepwise
 * entry points below folded into a callback; n candidates in, n records out, host pointers. */
typedef int (*pais_record_source_fn)(void *user, int n, const pais_candidate *cands, pais_patch_result *out, int has_seeds);
int  pais_mvs_set_record_source(pais_mvs *m, pais_record_source_fn fn, void *user);

/* FAILURE INJECTION into the sharded batch protocol on THIS rank (tests/test_distributed_cpu.py; drivers with a caller-supplied
 * all-gather -- the host instance of shard_submit / shard_finish).  The statuses are those the device path produces by itself:
 *   what 1  the next `count` sharded batches: this rank's header says PAIS_WIRE_RC_RING_RETRY (its k_pso_ring pass did not
 *           complete) -> every rank must take a second exchange, after this rank has refined its shard again;
 *   what 2  the `count`-th growth of the exchange buffers from now fails on this rank -> the growth handshake fails every rank;
 *   what 3  the `count`-th sharded refinement from now fails on this rank -> its header carries the status, every rank fails. */
int  pais_mvs_test_inject(pais_mvs *m, int what, int count);

#ifdef __cplusplus
}
#endif
#endif /* PAIS_TEST_HOOKS_H */
