//go:build biosketch

package sketches

/*
#cgo LDFLAGS: -lbiosketch
#include <stdlib.h>
#include "biosketch.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"
)

// ABIVersion is bsk_abi_version of the loaded library (the header this shim was written against: BSK_ABI_VERSION).
func ABIVersion() int { return int(C.bsk_abi_version()) }

// DeviceCount is the number of gfx950 devices the library sees (bsk_device_count).
func DeviceCount() (int, error) {
	var n C.int
	if rc := C.bsk_device_count(&n); rc != C.BSK_OK {
		return 0, fmt.Errorf("bsk_device_count: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return int(n), nil
}

// Sync waits for everything queued on the engine's stream (bsk_ctx_sync).
func (e *Engine) Sync() error { return e.err(C.bsk_ctx_sync(e.ctx)) }

// Info: sequences, bases, the longest sequence and the sequences with a non-ACGT letter of a batch (bsk_batch_info).
func (b *Batch) Info() (reads, bases, maxLen, nonACGT uint64) {
	var n, nb, ml, na C.uint64_t
	C.bsk_batch_info(b.h, &n, &nb, &ml, &na)
	runtime.KeepAlive(b)
	return uint64(n), uint64(nb), uint64(ml), uint64(na)
}

// ---- streaming: one batch object per goroutine, re-filled chunk after chunk (no allocation in steady state) ----------------

// Refill is NewBatch into an existing batch object (b may be nil the first time): bsk_batch_refill_ascii.  bytes / offsets are C
// memory (pinned memory makes the copy asynchronous); the old contents of b are gone after the call.
func (e *Engine) Refill(b *Batch, bytes unsafe.Pointer, offsets unsafe.Pointer, n int, protein bool) (*Batch, error) {
	if b == nil {
		b = &Batch{eng: e}
		runtime.SetFinalizer(b, func(b *Batch) { C.bsk_batch_destroy(b.h) })
	}
	alpha := C.int(C.BSK_ALPHA_DNA)
	if protein {
		alpha = C.BSK_ALPHA_PROTEIN
	}
	rc := C.bsk_batch_refill_ascii(e.ctx, &b.h, (*C.uint8_t)(bytes), (*C.uint64_t)(offsets), C.uint64_t(n), alpha)
	b.n, b.protein = n, protein
	return b, e.err(rc)
}

// RefillPacked: the same for reads the host already packed (2 bits per base, 16 bases per uint32, every read on a word boundary;
// desc[i] = first_word<<24 | bases): bsk_batch_refill_packed -- a quarter of the bytes over the link, no pack kernel.
func (e *Engine) RefillPacked(b *Batch, words unsafe.Pointer, nWords uint64, desc unsafe.Pointer, n int) (*Batch, error) {
	if b == nil {
		b = &Batch{eng: e}
		runtime.SetFinalizer(b, func(b *Batch) { C.bsk_batch_destroy(b.h) })
	}
	rc := C.bsk_batch_refill_packed(e.ctx, &b.h, (*C.uint32_t)(words), C.uint64_t(nWords), (*C.uint64_t)(desc), C.uint64_t(n))
	b.n, b.protein = n, false
	return b, e.err(rc)
}

// NewBatchPacked creates a batch from packed reads (bsk_batch_from_packed).
func (e *Engine) NewBatchPacked(words unsafe.Pointer, nWords uint64, desc unsafe.Pointer, n int) (*Batch, error) {
	b := &Batch{eng: e, n: n}
	rc := C.bsk_batch_from_packed(e.ctx, (*C.uint32_t)(words), C.uint64_t(nWords), (*C.uint64_t)(desc), C.uint64_t(n), &b.h)
	if err := e.err(rc); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(b, func(b *Batch) { C.bsk_batch_destroy(b.h) })
	return b, nil
}

// ---- device-side views of a result --------------------------------------------------------------------------------------

// Digest: the order-independent checksum of a result, its tuple count and the reads per status flag (bsk_result_digest).
func (r *DeviceResult) Digest() (checksum, tuples uint64, short, illegal, firstWindowTie, nonACGT uint64, err error) {
	var ck, nt C.uint64_t
	var sc [4]C.uint64_t
	rc := C.bsk_result_digest(r.eng.ctx, r.h, &ck, &nt, &sc[0])
	runtime.KeepAlive(r)
	return uint64(ck), uint64(nt), uint64(sc[0]), uint64(sc[1]), uint64(sc[2]), uint64(sc[3]), r.eng.err(rc)
}

// DevicePointers: the result's device arrays (bsk_result_device; refs nil for a wide result: see DeviceWide).  Valid until the
// result is released or sketched into again.
func (r *DeviceResult) DevicePointers() (refs, status, hash, pos unsafe.Pointer) {
	var rf, h *C.uint64_t
	var st *C.uint8_t
	var p *C.uint32_t
	C.bsk_result_device(r.h, &rf, &st, &h, &p)
	return unsafe.Pointer(rf), unsafe.Pointer(st), unsafe.Pointer(h), unsafe.Pointer(p)
}

// DeviceWide: first[] / count[] of a wide result (tiled long sequences: bsk_result_device_wide).
func (r *DeviceResult) DeviceWide() (first, count unsafe.Pointer) {
	var f, c *C.uint64_t
	C.bsk_result_device_wide(r.h, &f, &c)
	return unsafe.Pointer(f), unsafe.Pointer(c)
}

// DevicePointers of the sets: offsets[n_sets+1], values[] (bsk_sets_device).
func (s *Sets) DevicePointers() (offsets, values unsafe.Pointer) {
	var o, v *C.uint64_t
	C.bsk_sets_device(s.h, &o, &v)
	return unsafe.Pointer(o), unsafe.Pointer(v)
}

// ---- one process (or goroutine) per GPU: the per-rank form of the one collective ----------------------------------------------

// UniqueID is what rank 0 creates and hands to the other ranks by whatever channel the job has (bsk_comm_unique_id).
func UniqueID() ([C.BSK_UNIQUE_ID_BYTES]byte, error) {
	var id [C.BSK_UNIQUE_ID_BYTES]byte
	rc := C.bsk_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0])))
	if rc != C.BSK_OK {
		return id, fmt.Errorf("bsk_comm_unique_id: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return id, nil
}

// JoinRank joins the communicator of a job of `world` ranks (bsk_comm_init_rank).
func (e *Engine) JoinRank(id [C.BSK_UNIQUE_ID_BYTES]byte, rank, world int) error {
	return e.err(C.bsk_comm_init_rank(e.ctx, (*C.uint8_t)(unsafe.Pointer(&id[0])), C.int(rank), C.int(world)))
}

// GatherCounts all_gathers this rank's counters with every other rank's (bsk_gather_counts): world * len(mine) values in rank order.
func (e *Engine) GatherCounts(mine []uint64, world int) ([]uint64, error) {
	all := make([]uint64, world*len(mine))
	rc := C.bsk_gather_counts(e.ctx, (*C.uint64_t)(unsafe.Pointer(&mine[0])), C.int(len(mine)), (*C.uint64_t)(unsafe.Pointer(&all[0])))
	return all, e.err(rc)
}

// LeaveComm destroys the engine's communicator (bsk_comm_destroy).
func (e *Engine) LeaveComm() { C.bsk_comm_destroy(e.ctx) }

// ---- the single-device statistics forms -----------------------------------------------------------------------------------

// SketchFile: one file through one device's pipeline, statistics only (bsk_pipeline_fastx).
func SketchFile(device int, path string, p C.bsk_params, streams int, chunkRecords uint64, fetch bool) (PipelineStats, error) {
	cp := C.CString(path)
	defer C.free(unsafe.Pointer(cp))
	var st C.bsk_pipeline_stats
	f := C.int(0)
	if fetch {
		f = 1
	}
	rc := C.bsk_pipeline_fastx(C.int(device), cp, -1, &p, C.int(streams), C.uint64_t(chunkRecords), f, &st)
	if rc != C.BSK_OK {
		return statsFromC(&st), fmt.Errorf("bsk_pipeline_fastx: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return statsFromC(&st), nil
}

// SketchMemory: sequences in C host memory through one device's pipeline, statistics only (bsk_pipeline_memory).
func SketchMemory(device int, bytes unsafe.Pointer, offsets unsafe.Pointer, n uint64, alphabet int, p C.bsk_params, streams int, chunkRecords uint64, repeat int,
	fetch bool) (PipelineStats, error) {
	var st C.bsk_pipeline_stats
	f := C.int(0)
	if fetch {
		f = 1
	}
	rc := C.bsk_pipeline_memory(C.int(device), (*C.uint8_t)(bytes), (*C.uint64_t)(offsets), C.uint64_t(n), C.int(alphabet), &p, C.int(streams),
		C.uint64_t(chunkRecords), C.int(repeat), f, &st)
	if rc != C.BSK_OK {
		return statsFromC(&st), fmt.Errorf("bsk_pipeline_memory: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return statsFromC(&st), nil
}
