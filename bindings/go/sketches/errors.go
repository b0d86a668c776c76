//go:build biosketch

package sketches

import (
	"errors"
	"fmt"
)

// The reference's sentinel errors: same exported names, same texts (sketches/iterator.go:34-53, sketches/sketch.go:32-42).
// Callers compare them by identity, so the shim must own them -- this package REPLACES github.com/shenwei356/bio/sketches
// in a build (go.mod `replace`, see INTEGRATION.md), it does not sit beside it.  The C ABI reports them as codes 1..11
// (include/biosketch.h bsk_err); engine.go maps the codes back to these variables.
var (
	ErrInvalidK     = fmt.Errorf("sketches: invalid k-mer size")                                         // iterator.go:35
	ErrEmptySeq     = fmt.Errorf("sketches: empty sequence")                                             // iterator.go:38
	ErrShortSeq     = fmt.Errorf("sketches: sequence too short")                                         // iterator.go:41
	ErrIllegalBase  = errors.New("sketches: illegal base")                                               // iterator.go:44
	ErrKTooLarge    = fmt.Errorf("sketches: k-mer size is too large")                                    // iterator.go:47
	ErrInvalidM     = fmt.Errorf("sketches: invalid m-mer size, should be in range of [4, k]")           // iterator.go:50
	ErrInvalidScale = fmt.Errorf("sketches: invalid scale, should be in range of [1, k-m+1]")            // iterator.go:53
	ErrInvalidS     = fmt.Errorf("kmers: invalid s-mer size")                                            // sketch.go:33
	ErrInvalidW     = fmt.Errorf("kmers: invalid minimimzer window")                                     // sketch.go:36 (upstream's spelling)
	ErrBufNil       = fmt.Errorf("kmers: buffer slice is nil")                                           // sketch.go:39
	ErrBufNotEmpty  = fmt.Errorf("kmers: buffer has elements")                                           // sketch.go:42
)

// IdxValue is the (position, hash) pair the reference exports beside its sketches (sketch.go:496).  Result.IdxValues
// hands a record's tuples out in this form.
type IdxValue struct {
	Idx int    // index
	Val uint64 // hash
}
