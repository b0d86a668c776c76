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
	"unsafe"

	"github.com/shenwei356/bio/seq"
	"github.com/shenwei356/bio/seqio/fastx"
)

// sentinel errors: the SAME variables as the reference (iterator.go:34-53, sketch.go:32-42),
// so callers that compare by identity keep working.
var errByCode = map[C.int]error{
	C.BSK_ERR_INVALID_K:     ErrInvalidK,
	C.BSK_ERR_EMPTY_SEQ:     ErrEmptySeq,
	C.BSK_ERR_SHORT_SEQ:     ErrShortSeq,
	C.BSK_ERR_ILLEGAL_BASE:  ErrIllegalBase,
	C.BSK_ERR_K_TOO_LARGE:   ErrKTooLarge,
	C.BSK_ERR_INVALID_M:     ErrInvalidM,
	C.BSK_ERR_INVALID_SCALE: ErrInvalidScale,
	C.BSK_ERR_INVALID_S:     ErrInvalidS,
	C.BSK_ERR_INVALID_W:     ErrInvalidW,
	C.BSK_ERR_BUF_NIL:       ErrBufNil,
	C.BSK_ERR_BUF_NOT_EMPTY: ErrBufNotEmpty,
}

// Engine is one GPU context (bsk_ctx).  Use one per goroutine that drives a GPU.
type Engine struct{ ctx *C.bsk_ctx }

func NewEngine(device int) (*Engine, error) {
	e := &Engine{}
	if rc := C.bsk_ctx_create(C.int(device), &e.ctx); rc != C.BSK_OK {
		return nil, fmt.Errorf("biosketch: %s", C.GoString(C.bsk_err_name(rc)))
	}
	runtime.SetFinalizer(e, func(e *Engine) { C.bsk_ctx_destroy(e.ctx) })
	return e, nil
}

func (e *Engine) err(rc C.int) error {
	if rc == C.BSK_OK {
		return nil
	}
	if s, ok := errByCode[rc]; ok {
		return s
	}
	return errors.New("biosketch: " + C.GoString(C.bsk_err_name(rc)) + ": " + C.GoString(C.bsk_last_error(e.ctx)))
}

// Batch is a device-resident set of sequences (2-bit packed for DNA).
type Batch struct {
	eng     *Engine
	h       *C.bsk_batch
	n       int
	protein bool
}

// NewBatch copies the sequence bytes of the records into one C buffer (the fastx reader
// reuses its record buffer, seqio/fastx/reader.go:229-232) and hands it to the device.
// No Go pointer is retained by C after the call returns.
func (e *Engine) NewBatch(records []*fastx.Record) (*Batch, error) {
	seqs := make([]*seq.Seq, len(records))
	for i, r := range records {
		seqs[i] = r.Seq
	}
	return e.NewBatchFromSeqs(seqs)
}

func (e *Engine) NewBatchFromSeqs(seqs []*seq.Seq) (*Batch, error) {
	n := len(seqs)
	total := 0
	protein := n > 0 && seqs[0].Alphabet == seq.Protein // same pointer test as iterator-protein.go:62
	for _, s := range seqs {
		total += len(s.Seq)
	}
	bytes := (*[1 << 40]byte)(C.malloc(C.size_t(total + 1)))[: total+1 : total+1]
	offs := (*[1 << 37]C.uint64_t)(C.malloc(C.size_t(8 * (n + 1))))[: n+1 : n+1]
	defer C.free(unsafe.Pointer(&bytes[0]))
	defer C.free(unsafe.Pointer(&offs[0]))
	o := 0
	for i, s := range seqs {
		offs[i] = C.uint64_t(o)
		o += copy(bytes[o:], s.Seq)
	}
	offs[n] = C.uint64_t(o)
	// the batch's alphabet: the first sequence's, as a file has one (fastx guesses it once, reader.go:430-435).  Only the
	// two-strand NextKmer mode tells the nucleotide alphabets apart (RevComInplace pairs letters per alphabet, iterator.go:719).
	alpha := C.int(C.BSK_ALPHA_DNA)
	if n > 0 {
		switch seqs[0].Alphabet {
		case seq.Protein:
			alpha = C.BSK_ALPHA_PROTEIN
		case seq.DNA:
			alpha = C.BSK_ALPHA_DNA_PLAIN
		case seq.RNA:
			alpha = C.BSK_ALPHA_RNA
		case seq.RNAredundant:
			alpha = C.BSK_ALPHA_RNA_REDUNDANT
		case seq.Unlimit:
			alpha = C.BSK_ALPHA_UNLIMIT
		}
	}
	b := &Batch{eng: e, n: n, protein: protein}
	rc := C.bsk_batch_from_ascii(e.ctx, (*C.uint8_t)(unsafe.Pointer(&bytes[0])), &offs[0], C.uint64_t(n), alpha, &b.h)
	if err := e.err(rc); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(b, func(b *Batch) { C.bsk_batch_destroy(b.h) })
	return b, nil
}

// Result holds the tuples of one launch on the host (CSR by record).
type Result struct {
	kind    C.int
	offsets []uint64
	status  []uint8
	hash    []uint64
	pos     []uint32 // nil for the "every position" kinds
	seqLen  []int
	k       int
}

func (b *Batch) run(p C.bsk_params) (*Result, error) {
	var r *C.bsk_result
	rc0 := C.bsk_sketch(b.eng.ctx, b.h, &p, &r)
	runtime.KeepAlive(b)
	if err := b.eng.err(rc0); err != nil {
		return nil, err
	}
	defer C.bsk_result_release(r)
	var nReads, nTuples C.uint64_t
	var hasPos C.int
	C.bsk_result_info(r, &nReads, &nTuples, &hasPos)
	res := &Result{kind: C.int(p.kind), k: int(p.k)}
	res.offsets = make([]uint64, int(nReads)+1)
	res.status = make([]uint8, int(nReads)+1)
	res.hash = make([]uint64, int(nTuples)+1)
	var posPtr *C.uint32_t
	if hasPos != 0 {
		res.pos = make([]uint32, int(nTuples)+1)
		posPtr = (*C.uint32_t)(unsafe.Pointer(&res.pos[0]))
	}
	// Go memory is only written during this call (cgo pointer rules)
	rc := C.bsk_result_fetch(b.eng.ctx, r, 0, nReads, (*C.uint64_t)(unsafe.Pointer(&res.offsets[0])),
		(*C.uint8_t)(unsafe.Pointer(&res.status[0])), (*C.uint64_t)(unsafe.Pointer(&res.hash[0])), posPtr, nTuples+1)
	if err := b.eng.err(rc); err != nil {
		return nil, err
	}
	return res, nil
}

// SketchSets runs the sketch named by p and returns, per sequence (wholeBatch = false) or for the whole batch, the
// sorted distinct hash values -- optionally FracMinHash-filtered (hash <= MaxUint64/scale) -- computed on the device
// (bsk_result_sets): what a caller otherwise builds with append + sort + dedup over the Next() stream.
func (b *Batch) sketchSets(p C.bsk_params, wholeBatch bool, scale int) (offsets []uint64, values []uint64, err error) {
	var r *C.bsk_result
	if err = b.eng.err(C.bsk_sketch(b.eng.ctx, b.h, &p, &r)); err != nil {
		return nil, nil, err
	}
	defer C.bsk_result_release(r)
	scope := C.int(C.BSK_SETS_PER_SEQUENCE)
	if wholeBatch {
		scope = C.int(C.BSK_SETS_WHOLE_BATCH)
	}
	var s *C.bsk_sets
	if err = b.eng.err(C.bsk_result_sets(b.eng.ctx, r, scope, C.int(scale), &s)); err != nil {
		return nil, nil, err
	}
	defer C.bsk_sets_release(s)
	var nSets, nValues C.uint64_t
	C.bsk_sets_info(s, &nSets, &nValues)
	offsets = make([]uint64, int(nSets)+1)
	values = make([]uint64, int(nValues)+1)
	rc := C.bsk_sets_fetch(b.eng.ctx, s, 0, nSets, (*C.uint64_t)(unsafe.Pointer(&offsets[0])), (*C.uint64_t)(unsafe.Pointer(&values[0])), nValues+1)
	if err = b.eng.err(rc); err != nil {
		return nil, nil, err
	}
	return offsets, values[:int(nValues)], nil
}

// MinimizerSets: sorted distinct minimizer hashes of every sequence (or of the batch).
func (b *Batch) MinimizerSets(k, w int, wholeBatch bool, scale int) ([]uint64, []uint64, error) {
	return b.sketchSets(C.bsk_params{kind: C.BSK_MINIMIZER, k: C.int32_t(k), w: C.int32_t(w), canonical: 1}, wholeBatch, scale)
}

// SyncmerSets: the same for the syncmer sketch.
func (b *Batch) SyncmerSets(k, s int, wholeBatch bool, scale int) ([]uint64, []uint64, error) {
	return b.sketchSets(C.bsk_params{kind: C.BSK_SYNCMER, k: C.int32_t(k), s: C.int32_t(s), canonical: 1}, wholeBatch, scale)
}

func b2i(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

// one launch per constructor of the reference
func (b *Batch) HashIterators(k int, canonical, circular bool) (*Result, error) {
	return b.run(C.bsk_params{kind: C.BSK_NTHASH, k: C.int32_t(k), canonical: b2i(canonical), circular: b2i(circular)})
}
func (b *Batch) KmerIterators(k int, canonical, circular bool) (*Result, error) {
	return b.run(C.bsk_params{kind: C.BSK_KMER, k: C.int32_t(k), canonical: b2i(canonical), circular: b2i(circular)})
}
func (b *Batch) SimHashIterators(k, m, scale int, canonical, circular bool) (*Result, error) {
	return b.run(C.bsk_params{kind: C.BSK_SIMHASH, k: C.int32_t(k), m: C.int32_t(m), scale: C.int32_t(scale),
		canonical: b2i(canonical), circular: b2i(circular)})
}
func (b *Batch) MinimizerSketches(k, w int, circular bool) (*Result, error) {
	return b.run(C.bsk_params{kind: C.BSK_MINIMIZER, k: C.int32_t(k), w: C.int32_t(w), canonical: 1, circular: b2i(circular)})
}
func (b *Batch) SyncmerSketches(k, s int, circular bool) (*Result, error) {
	return b.run(C.bsk_params{kind: C.BSK_SYNCMER, k: C.int32_t(k), s: C.int32_t(s), canonical: 1, circular: b2i(circular)})
}
func (b *Batch) ProteinIterators(k, codonTable, frame int) (*Result, error) {
	return b.run(C.bsk_params{kind: C.BSK_PROT_HASH, k: C.int32_t(k), codon_table: C.int32_t(codonTable), frame: C.int32_t(frame)})
}
func (b *Batch) ProteinMinimizerSketches(k, codonTable, frame, w int) (*Result, error) {
	return b.run(C.bsk_params{kind: C.BSK_PROT_MINIMIZER, k: C.int32_t(k), w: C.int32_t(w), codon_table: C.int32_t(codonTable),
		frame: C.int32_t(frame)})
}

// Translate is (*seq.Seq).Translate(codonTable, frame, false, false, true, false) (seq/seq.go:685) for every sequence of a
// DNA/RNA batch, on the device; the protein constructors above do this themselves when the batch is not protein.
func (b *Batch) Translate(codonTable, frame int) (*Batch, error) {
	t := &Batch{eng: b.eng, n: b.n, protein: true}
	rc := C.bsk_batch_translate(b.eng.ctx, b.h, C.int(codonTable), C.int(frame), &t.h)
	runtime.KeepAlive(b) // the source batch must outlive the call (its finalizer frees device memory)
	if err := b.eng.err(rc); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(t, func(t *Batch) { C.bsk_batch_destroy(t.h) })
	return t, nil
}

func (r *Result) slice(i int) (codes []uint64, pos []uint32, status uint8, err error) {
	a, e := r.offsets[i], r.offsets[i+1]
	status = r.status[i]
	if status&C.BSK_ST_CODE_MASK == C.BSK_ST_SHORT {
		return nil, nil, status, ErrShortSeq // what the reference constructor returns for this record
	}
	codes = r.hash[a:e]
	if r.pos != nil {
		pos = r.pos[a:e]
	}
	return
}

// ---- multi-GPU: one Engine per GPU, reads sharded by record, ONE collective at the end (DESIGN.md 5) ----

// JoinEngines forms the RCCL communicator of the engines of one process (bsk_comm_init_all): rank i = engines[i].
func JoinEngines(engines []*Engine) error {
	ctxs := make([]*C.bsk_ctx, len(engines))
	for i, e := range engines {
		ctxs[i] = e.ctx
	}
	rc := C.bsk_comm_init_all((**C.bsk_ctx)(unsafe.Pointer(&ctxs[0])), C.int(len(engines)))
	runtime.KeepAlive(engines)
	return engines[0].err(rc)
}

// GatherCounts all_gathers every engine's counters (reads, bases, tuples, flagged reads ...) over RCCL / xGMI
// (bsk_gather_counts_all) and returns them in rank order.  Call it once the worker goroutines have joined.
func GatherCounts(engines []*Engine, mine [][]uint64) ([][]uint64, error) {
	n, nc := len(engines), len(mine[0])
	ctxs := make([]*C.bsk_ctx, n)
	flat := make([]uint64, n*nc)
	for i, e := range engines {
		ctxs[i] = e.ctx
		copy(flat[i*nc:], mine[i])
	}
	all := make([]uint64, n*nc)
	rc := C.bsk_gather_counts_all((**C.bsk_ctx)(unsafe.Pointer(&ctxs[0])), C.int(n), (*C.uint64_t)(unsafe.Pointer(&flat[0])), C.int(nc),
		(*C.uint64_t)(unsafe.Pointer(&all[0])))
	runtime.KeepAlive(engines)
	if err := engines[0].err(rc); err != nil {
		return nil, err
	}
	out := make([][]uint64, n)
	for i := range out {
		out[i] = all[i*nc : (i+1)*nc]
	}
	return out, nil
}

// ---- file -> tuples, stages overlapped (bsk_pipeline_fastx_files; DESIGN.md 4) ----

// PipelineStats mirrors bsk_pipeline_stats: whole-job counters and where the time went.
type PipelineStats struct {
	Records, Bases, Tuples, Chunks, Checksum uint64
	Seconds, ReaderSeconds, ReaderWaitSeconds float64
	H2DPackSeconds, KernelSeconds, FetchSeconds float64
	Streams, ReaderThreads                      int
	ReparsedPieces                              uint64
	PinSeconds                                  float64
}

// SketchFiles runs FASTA/FASTQ files (plain, BGZF or gzip) through one device pipeline: `readers` files are read at the same
// time (0: up to eight), plain and BGZF files by the block-parallel reader, the sketch named by p runs on `streams` HIP streams and
// every tuple comes back to pinned host memory.  What a host loop over fastx.Reader + NewMinimizerSketch per record becomes when
// the GPU does the hashing: the reader is no longer allowed to be one goroutine.
func SketchFiles(device int, paths []string, p C.bsk_params, streams, readers int, chunkRecords uint64) (PipelineStats, error) {
	cs := make([]*C.char, len(paths))
	for i, s := range paths {
		cs[i] = C.CString(s)
		defer C.free(unsafe.Pointer(cs[i]))
	}
	var st C.bsk_pipeline_stats
	rc := C.bsk_pipeline_fastx_files(C.int(device), (**C.char)(unsafe.Pointer(&cs[0])), C.int(len(paths)), -1, &p, C.int(streams), C.int(readers),
		C.uint64_t(chunkRecords), 1, &st)
	out := PipelineStats{uint64(st.records), uint64(st.bases), uint64(st.tuples), uint64(st.chunks), uint64(st.checksum),
		float64(st.seconds), float64(st.reader_seconds), float64(st.reader_wait_seconds),
		float64(st.h2d_pack_seconds), float64(st.kernel_seconds), float64(st.fetch_seconds),
		int(st.n_streams), int(st.reader_threads), uint64(st.reparsed_pieces), float64(st.pin_seconds)}
	if rc != C.BSK_OK {
		return out, fmt.Errorf("bsk_pipeline_fastx_files: %s", C.GoString(C.bsk_err_name(rc)))
	}
	return out, nil
}

// PipelineTrim returns the pinned host buffers SketchFiles keeps pooled between calls (bsk_pipeline_trim).
func PipelineTrim() { C.bsk_pipeline_trim() }
