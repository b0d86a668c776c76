//go:build biosketch

package sketches

/*
#cgo LDFLAGS: -lbiosketch
#include <stdlib.h>
#include "biosketch.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"sync"
	"unsafe"
)

// ---- the pipeline with a consumer: what fastx's ChunkChan is to the reference's callers ----------------------------------
//
// seqio/fastx/reader.go:562-608 hands every chunk of records, in input order, to whoever ranges over the channel; a worker
// then loops NewMinimizerSketch / Next over the records.  Here the device does that loop: Chunks() delivers, in input order,
// every chunk's sketches (or their on-device reduction to sorted distinct FracMinHash sets).

// Sink selects what crosses the link for every chunk (BSK_SINK_*).
type Sink int

const (
	SinkCounts Sink = C.BSK_SINK_COUNTS // counts + the device-side digest only
	SinkTuples Sink = C.BSK_SINK_TUPLES // every (hash, position, strand) tuple, in Next() order
	SinkSets   Sink = C.BSK_SINK_SETS   // per record: ascending distinct hashes with hash <= MaxUint64/scale (iterator.go:181-185)
)

// PipelineConfig mirrors bsk_pipeline_config.
type PipelineConfig struct {
	Devices      []int  // the GPUs of the run; a device may be named more than once
	Streams      int    // workers (context + HIP stream) per device
	ChunkRecords uint64 // records per chunk (0: the reader's default)
	Sink         Sink
	SetsScale    int // SinkSets: FracMinHash scale (<= 1: no filter)
	Alphabet     int // bsk_alphabet, -1: guess from the first record
	Readers      int // several files: how many are read at once (0: min(len(paths), 8))
}

// Chunk is one delivered chunk.  Its slices alias pinned memory of the pipeline and are valid until the next receive from
// Chunks() (or Close): copy what must outlive that.  Record i owns Hash[Offsets[i]:Offsets[i+1]].
type Chunk struct {
	Sequence    uint64 // 0, 1, 2 ...: the delivery order
	SourceIndex int    // which of the run's files
	Device      int
	FirstRecord uint64 // index of the chunk's first record in its source
	Records     uint64
	Bases       uint64
	Tuples      uint64
	Checksum    uint64
	LinkBytes   uint64
	Offsets     []uint64
	Status      []uint8
	Hash        []uint64
	Pos         []uint32 // SinkTuples, kinds with positions: bit 31 = strand (BSK_POS_STRAND_BIT); nil otherwise
}

// Pipeline is a running bsk_pipeline.
type Pipeline struct {
	h     *C.bsk_pipeline
	devs  *C.int
	held  *C.bsk_chunk
	Stats PipelineStats // filled by Close

	mu   sync.Mutex     // guards err (written by the goroutine of Chunks, read by Err)
	err  error
	done chan struct{}  // closed by Close: the goroutine of Chunks stops
	wg   sync.WaitGroup // the goroutine of Chunks; Close waits for it before the pipeline is freed
}

func (pl *Pipeline) setErr(err error) {
	pl.mu.Lock()
	if pl.err == nil {
		pl.err = err
	}
	pl.mu.Unlock()
}

func (cfg *PipelineConfig) c() (C.bsk_pipeline_config, *C.int) {
	n := len(cfg.Devices)
	devs := (*C.int)(C.malloc(C.size_t(4 * (n + 1))))
	dv := (*[1 << 20]C.int)(unsafe.Pointer(devs))[:n:n]
	for i, d := range cfg.Devices {
		dv[i] = C.int(d)
	}
	return C.bsk_pipeline_config{devices: devs, n_devices: C.int32_t(n), n_streams: C.int32_t(cfg.Streams), chunk_records: C.uint64_t(cfg.ChunkRecords),
		sink: C.int32_t(cfg.Sink), sets_scale: C.int32_t(cfg.SetsScale), alphabet: C.int32_t(cfg.Alphabet), n_readers: C.int32_t(cfg.Readers)}, devs
}

// OpenFiles starts a pipeline over FASTA/FASTQ files (plain, BGZF or gzip): bsk_pipeline_open_fastx.
func OpenFiles(cfg PipelineConfig, paths []string, p C.bsk_params) (*Pipeline, error) {
	if len(paths) == 0 || len(cfg.Devices) == 0 {
		return nil, errors.New("biosketch: OpenFiles needs paths and devices")
	}
	cc, devs := cfg.c()
	cs := make([]*C.char, len(paths))
	for i, s := range paths {
		cs[i] = C.CString(s)
		defer C.free(unsafe.Pointer(cs[i]))
	}
	pl := &Pipeline{devs: devs, done: make(chan struct{})}
	rc := C.bsk_pipeline_open_fastx(&cc, (**C.char)(unsafe.Pointer(&cs[0])), C.int(len(paths)), &p, &pl.h)
	if rc != C.BSK_OK {
		C.free(unsafe.Pointer(devs))
		return nil, fmt.Errorf("bsk_pipeline_open_fastx: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return pl, nil
}

// OpenMemory starts a pipeline over sequences already in (C-allocated or pinned) host memory: bsk_pipeline_open_memory.
// bytes / offsets must stay valid and unchanged until Close -- pass C memory, not Go slices (cgo pointer rules).
func OpenMemory(cfg PipelineConfig, bytes unsafe.Pointer, offsets unsafe.Pointer, n uint64, repeat int, p C.bsk_params) (*Pipeline, error) {
	cc, devs := cfg.c()
	pl := &Pipeline{devs: devs, done: make(chan struct{})}
	rc := C.bsk_pipeline_open_memory(&cc, (*C.uint8_t)(bytes), (*C.uint64_t)(offsets), C.uint64_t(n), C.int(repeat), &p, &pl.h)
	if rc != C.BSK_OK {
		C.free(unsafe.Pointer(devs))
		return nil, fmt.Errorf("bsk_pipeline_open_memory: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return pl, nil
}

// Next returns the next chunk in input order, nil at the end (bsk_pipeline_next); the chunk returned before is released.
func (pl *Pipeline) Next() (*Chunk, error) {
	if pl.held != nil {
		C.bsk_pipeline_release(pl.h, pl.held)
		pl.held = nil
	}
	out, c, err := pl.fetch()
	pl.held = c
	return out, err
}

// fetch takes the next chunk WITHOUT releasing any other (the caller owns the release of raw): what Next and Chunks share.
func (pl *Pipeline) fetch() (out *Chunk, raw *C.bsk_chunk, err error) {
	var c *C.bsk_chunk
	if rc := C.bsk_pipeline_next(pl.h, &c); rc != C.BSK_OK {
		if rc == -1 { // stopped by Close / bsk_pipeline_cancel: the end of the stream, not an error of the run
			return nil, nil, nil
		}
		err = fmt.Errorf("bsk_pipeline_next: %s: %s", C.GoString(C.bsk_err_name(rc)), C.GoString(C.bsk_pipeline_error(pl.h)))
		pl.setErr(err)
		return nil, nil, err
	}
	if c == nil {
		return nil, nil, nil
	}
	n, nv := int(c.n_records), int(c.n_values)
	out = &Chunk{Sequence: uint64(c.sequence), SourceIndex: int(c.source_index), Device: int(c.device), FirstRecord: uint64(c.first_record),
		Records: uint64(c.n_records), Bases: uint64(c.n_bases), Tuples: uint64(c.n_tuples), Checksum: uint64(c.checksum), LinkBytes: uint64(c.link_bytes)}
	if c.status != nil && n > 0 {
		out.Status = (*[1 << 40]uint8)(unsafe.Pointer(c.status))[:n:n]
	}
	// offsets and positions arrive narrow (u32 / u16) unless a read of 32 768 bases or more is in the chunk: widened here, once per chunk
	out.Offsets = make([]uint64, n+1)
	if c.offsets32 != nil {
		o := (*[1 << 38]uint32)(unsafe.Pointer(c.offsets32))[: n+1 : n+1]
		for i, v := range o {
			out.Offsets[i] = uint64(v)
		}
	} else if c.offsets64 != nil {
		copy(out.Offsets, (*[1 << 37]uint64)(unsafe.Pointer(c.offsets64))[:n+1:n+1])
	}
	if c.hash != nil && nv > 0 {
		out.Hash = (*[1 << 37]uint64)(unsafe.Pointer(c.hash))[:nv:nv]
	}
	if c.pos16 != nil && nv > 0 {
		p16 := (*[1 << 39]uint16)(unsafe.Pointer(c.pos16))[:nv:nv]
		out.Pos = make([]uint32, nv)
		for i, v := range p16 {
			out.Pos[i] = uint32(v&C.BSK_POS16_MASK) | uint32(v>>15)<<31
		}
	} else if c.pos32 != nil && nv > 0 {
		out.Pos = (*[1 << 38]uint32)(unsafe.Pointer(c.pos32))[:nv:nv]
	}
	return out, c, nil
}

// Chunks mirrors fastx.Reader.ChunkChan (seqio/fastx/reader.go:562-608): a channel of chunks in input order, closed at the end of the
// input, on error (see Err) or by Close.  A chunk is valid until the NEXT one is received: Hash, Status and Pos alias pinned memory of
// the pipeline, so the goroutine keeps the chunk the consumer is reading and gives it back only after the following chunk has been
// handed over (two of the pool's 2 * workers + 2 output buffers are held at most).  Call Chunks once per pipeline and do not mix it
// with Next.
func (pl *Pipeline) Chunks() <-chan *Chunk {
	ch := make(chan *Chunk)
	pl.wg.Add(1)
	go func() {
		defer pl.wg.Done()
		defer close(ch)
		var prev *C.bsk_chunk // the chunk the consumer received last and may still be reading
		defer func() {
			if prev != nil {
				C.bsk_pipeline_release(pl.h, prev)
			}
		}()
		for {
			select {
			case <-pl.done:
				return
			default:
			}
			c, raw, err := pl.fetch() // takes a buffer of its own; prev stays untouched
			if err != nil || c == nil {
				return
			}
			select {
			case ch <- c:
			case <-pl.done:
				C.bsk_pipeline_release(pl.h, raw)
				return
			}
			if prev != nil { // the consumer has taken c: what it held before is no longer its to read
				C.bsk_pipeline_release(pl.h, prev)
			}
			prev = raw
		}
	}()
	return ch
}

// Err is the error that ended Chunks() early, if any.
func (pl *Pipeline) Err() error {
	pl.mu.Lock()
	defer pl.mu.Unlock()
	return pl.err
}

// Close stops the run (if it has not ended) and frees the pipeline; Stats holds the run's counters afterwards.
func (pl *Pipeline) Close() error {
	if pl.h == nil {
		return nil
	}
	// a goroutine of Chunks may sit inside bsk_pipeline_next: wake it (cancel frees nothing), wait for it, and only then free the pipeline
	C.bsk_pipeline_cancel(pl.h)
	select {
	case <-pl.done:
	default:
		close(pl.done)
	}
	pl.wg.Wait()
	if pl.held != nil {
		C.bsk_pipeline_release(pl.h, pl.held)
		pl.held = nil
	}
	var st C.bsk_pipeline_stats
	rc := C.bsk_pipeline_close(pl.h, &st)
	pl.h = nil
	C.free(unsafe.Pointer(pl.devs))
	pl.Stats = statsFromC(&st)
	if rc != C.BSK_OK && rc != -1 { // -1: closed before the end of the input
		return fmt.Errorf("bsk_pipeline_close: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return nil
}

func statsFromC(st *C.bsk_pipeline_stats) PipelineStats {
	return PipelineStats{uint64(st.records), uint64(st.bases), uint64(st.tuples), uint64(st.chunks), uint64(st.checksum),
		float64(st.seconds), float64(st.reader_seconds), float64(st.reader_wait_seconds),
		float64(st.h2d_pack_seconds), float64(st.kernel_seconds), float64(st.fetch_seconds),
		int(st.n_streams), int(st.reader_threads), uint64(st.reparsed_pieces), float64(st.pin_seconds)}
}

// SketchFilesMulti is SketchFiles over several GPUs of the node, one file: bsk_pipeline_fastx_multi (one chunk queue, `streams`
// workers per device; statistics only).
func SketchFilesMulti(devices []int, path string, p C.bsk_params, streams int, chunkRecords uint64) (PipelineStats, error) {
	dv := make([]C.int, len(devices))
	for i, d := range devices {
		dv[i] = C.int(d)
	}
	cp := C.CString(path)
	defer C.free(unsafe.Pointer(cp))
	var st C.bsk_pipeline_stats
	rc := C.bsk_pipeline_fastx_multi(&dv[0], C.int(len(dv)), cp, -1, &p, C.int(streams), C.uint64_t(chunkRecords), 1, &st)
	if rc != C.BSK_OK {
		return statsFromC(&st), fmt.Errorf("bsk_pipeline_fastx_multi: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return statsFromC(&st), nil
}

// SketchMemoryMulti: the same over sequences in C host memory (bsk_pipeline_memory_multi).
func SketchMemoryMulti(devices []int, bytes unsafe.Pointer, offsets unsafe.Pointer, n uint64, alphabet int, p C.bsk_params, streams int, chunkRecords uint64,
	repeat int) (PipelineStats, error) {
	dv := make([]C.int, len(devices))
	for i, d := range devices {
		dv[i] = C.int(d)
	}
	var st C.bsk_pipeline_stats
	rc := C.bsk_pipeline_memory_multi(&dv[0], C.int(len(dv)), (*C.uint8_t)(bytes), (*C.uint64_t)(offsets), C.uint64_t(n), C.int(alphabet), &p, C.int(streams),
		C.uint64_t(chunkRecords), C.int(repeat), 1, &st)
	if rc != C.BSK_OK {
		return statsFromC(&st), fmt.Errorf("bsk_pipeline_memory_multi: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return statsFromC(&st), nil
}

// ---- device-resident results for Go hosts that keep working on the device ------------------------------------------------

// DeviceResult keeps the result of one launch on the device (bsk_result): for callers that consume tuples there (sets, dense
// copies) or fetch ranges of records instead of everything.
type DeviceResult struct {
	eng *Engine
	h   *C.bsk_result
}

// Sketch runs p over the batch and leaves the result on the device.
func (b *Batch) Sketch(p C.bsk_params) (*DeviceResult, error) {
	r := &DeviceResult{eng: b.eng}
	rc := C.bsk_sketch(b.eng.ctx, b.h, &p, &r.h)
	runtime.KeepAlive(b)
	if err := b.eng.err(rc); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(r, func(r *DeviceResult) { C.bsk_result_release(r.h) })
	return r, nil
}

// Prepare does now what the first Sketch with p would do once per batch (length-binned units of a ragged batch) and returns
// the device milliseconds of the pass: bsk_batch_prepare.
func (b *Batch) Prepare(p C.bsk_params) (float32, error) {
	var ms C.float
	rc := C.bsk_batch_prepare(b.eng.ctx, b.h, &p, &ms)
	runtime.KeepAlive(b)
	return float32(ms), b.eng.err(rc)
}

// Plan names the kernels that ran (bsk_result_plan) and the class plan, if the batch was cut by length (bsk_result_class_plan).
func (r *DeviceResult) Plan() (kernel string, grid, wavesPerCU, classParts int, cutMs float32) {
	var k *C.char
	var g, w, np C.int
	var ms C.float
	C.bsk_result_plan(r.h, &k, &g, &w)
	C.bsk_result_class_plan(r.h, &np, &ms)
	return C.GoString(k), int(g), int(w), int(np), float32(ms)
}

// FetchNarrow copies records [first, first+count) to the host with u32 offsets and u16 positions (bit 15 = strand):
// bsk_result_fetch_narrow -- 10 bytes per tuple + 5 per record over the link instead of 12 + 17.
func (r *DeviceResult) FetchNarrow(first, count uint64) (offsets []uint32, status []uint8, hash []uint64, pos []uint16, err error) {
	var nReads, nTuples C.uint64_t
	var hasPos C.int
	C.bsk_result_info(r.h, &nReads, &nTuples, &hasPos)
	offsets = make([]uint32, count+1)
	status = make([]uint8, count+1)
	hash = make([]uint64, uint64(nTuples)+1)
	var pp *C.uint16_t
	if hasPos != 0 {
		pos = make([]uint16, uint64(nTuples)+1)
		pp = (*C.uint16_t)(unsafe.Pointer(&pos[0]))
	}
	var got C.uint64_t
	rc := C.bsk_result_fetch_narrow(r.eng.ctx, r.h, C.uint64_t(first), C.uint64_t(count), (*C.uint32_t)(unsafe.Pointer(&offsets[0])),
		(*C.uint8_t)(unsafe.Pointer(&status[0])), (*C.uint64_t)(unsafe.Pointer(&hash[0])), pp, nTuples+1, &got)
	runtime.KeepAlive(r)
	if err = r.eng.err(rc); err != nil {
		return nil, nil, nil, nil, err
	}
	hash = hash[:got]
	if pos != nil {
		pos = pos[:got]
	}
	return offsets, status[:count], hash, pos, nil
}

// FetchStatus copies the status bytes of records [first, first+count): bsk_result_fetch_status.
func (r *DeviceResult) FetchStatus(first, count uint64) ([]uint8, error) {
	st := make([]uint8, count+1)
	rc := C.bsk_result_fetch_status(r.eng.ctx, r.h, C.uint64_t(first), C.uint64_t(count), (*C.uint8_t)(unsafe.Pointer(&st[0])))
	runtime.KeepAlive(r)
	return st[:count], r.eng.err(rc)
}

// Compact leaves a dense CSR copy of the result on the device (bsk_result_compact) and returns its device pointers and size:
// what a device-side consumer (another kernel, a cgo library) reads.  Valid until the next Compact on the engine.
func (r *DeviceResult) Compact() (offsets, hash, pos unsafe.Pointer, tuples uint64, err error) {
	var o, h *C.uint64_t
	var p *C.uint32_t
	var n C.uint64_t
	rc := C.bsk_result_compact(r.eng.ctx, r.h, &o, &h, &p, &n)
	runtime.KeepAlive(r)
	return unsafe.Pointer(o), unsafe.Pointer(h), unsafe.Pointer(p), uint64(n), r.eng.err(rc)
}

// Sets is a device-resident collection of sorted distinct hash sets (bsk_sets), re-usable from chunk to chunk.
type Sets struct {
	eng *Engine
	h   *C.bsk_sets
}

// SetsInto reduces the result to per-record (or whole-batch) sets into s (nil the first time): bsk_result_sets_reuse -- the
// device arrays of s are kept and only grow, a streaming caller allocates nothing in steady state.
func (r *DeviceResult) SetsInto(s *Sets, wholeBatch bool, scale int) (*Sets, error) {
	if s == nil {
		s = &Sets{eng: r.eng}
		runtime.SetFinalizer(s, func(s *Sets) { C.bsk_sets_release(s.h) })
	}
	scope := C.int(C.BSK_SETS_PER_SEQUENCE)
	if wholeBatch {
		scope = C.int(C.BSK_SETS_WHOLE_BATCH)
	}
	rc := C.bsk_result_sets_reuse(r.eng.ctx, r.h, scope, C.int(scale), &s.h)
	runtime.KeepAlive(r)
	return s, r.eng.err(rc)
}

// Fetch copies all sets to the host in narrow form (u32 offsets): bsk_sets_fetch_narrow.
func (s *Sets) Fetch() (offsets []uint32, values []uint64, err error) {
	var nSets, nValues C.uint64_t
	C.bsk_sets_info(s.h, &nSets, &nValues)
	offsets = make([]uint32, uint64(nSets)+1)
	values = make([]uint64, uint64(nValues)+1)
	rc := C.bsk_sets_fetch_narrow(s.eng.ctx, s.h, (*C.uint32_t)(unsafe.Pointer(&offsets[0])), (*C.uint64_t)(unsafe.Pointer(&values[0])), nValues+1)
	runtime.KeepAlive(s)
	return offsets, values[:nValues], s.eng.err(rc)
}

// RunPipeline is the callback form (bsk_pipeline_run): fn sees every chunk in input order; a non-nil error stops the run.
func (pl *Pipeline) RunPipeline(fn func(*Chunk) error) error {
	for {
		c, err := pl.Next()
		if err != nil {
			pl.Close()
			return err
		}
		if c == nil {
			return pl.Close()
		}
		if err := fn(c); err != nil {
			pl.Close()
			return err
		}
	}
}
