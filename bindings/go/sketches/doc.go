// Package sketches is the cgo shim that puts the MI355X sketching engine
// (libbiosketch.so, include/biosketch.h) behind the API of
// github.com/shenwei356/bio/sketches.
//
// SOURCE ONLY: this image has no Go toolchain, so the files in this directory are
// not compiled or tested here (the tested boundary is the C ABI, through
// bio_amd/sketches.py and bio_amd/csrc/sketches.hpp).  A maintainer drops the
// directory next to the reference package, builds with
//
//	CGO_CFLAGS="-I$REPO/include" CGO_LDFLAGS="-L$REPO/bio_amd/csrc -lbiosketch" go build -tags biosketch
//
// and keeps every call site unchanged.  The reference's types become cursors over
// one record's slice of a batch result:
//
//	eng, _ := sketches.NewEngine(0)
//	batch, _ := eng.NewBatch(records)                       // []*fastx.Record, bytes are copied (reader.go:229-232)
//	res, _ := batch.MinimizerSketches(21, 11, false)        // one GPU launch for the whole batch
//	for i := range records {
//		sk, err := res.Sketch(i)                            // *sketches.Sketch, err == ErrShortSeq where upstream's would be
//		for { code, ok := sk.NextMinimizer(); if !ok { break }; _ = sk.Index() }
//	}
//
// The single-sequence constructors (NewMinimizerSketch, ...) are kept for source
// compatibility; they run a batch of one.
package sketches
