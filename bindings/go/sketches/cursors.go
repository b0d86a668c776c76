//go:build biosketch

package sketches

import "github.com/shenwei356/bio/seq"

// The four iterator types of the reference, as cursors over one record's tuples.
// Method sets and semantics are the reference's (file:line in comments).

// Iterator replaces sketches.Iterator (iterator.go:60-106).
type Iterator struct {
	codes   []uint64
	i, idx  int
	kmer    bool
	illegal bool
	perStr  int // non-canonical k-mer mode: codes per strand (Index() restarts, iterator.go:720)
}

func (r *Result) Iterator(i int) (*Iterator, error) {
	codes, _, st, err := r.slice(i)
	if err != nil {
		return nil, err
	}
	return &Iterator{codes: codes, idx: -1, kmer: r.kind == 1, illegal: st&0x0f == 0x02}, nil
}
func (it *Iterator) next() (uint64, bool) {
	if it.i >= len(it.codes) {
		return 0, false
	}
	c := it.codes[it.i]
	it.idx = it.i
	if it.perStr > 0 {
		it.idx = it.i % it.perStr
	}
	it.i++
	return c, true
}
func (it *Iterator) NextHash() (code uint64, ok bool) { return it.next() } // iterator.go:658
func (it *Iterator) NextSimHash() (code uint64, ok bool) { return it.next() } // iterator.go:191
func (it *Iterator) NextKmer() (uint64, bool, error) { // iterator.go:708
	c, ok := it.next()
	if !ok && it.illegal {
		return 0, false, ErrIllegalBase
	}
	return c, ok, nil
}
func (it *Iterator) Next() (uint64, bool, error) { // iterator.go:762
	if it.kmer {
		return it.NextKmer()
	}
	c, ok := it.next()
	return c, ok, nil
}
func (it *Iterator) Index() int { return it.idx } // iterator.go:776

// Sketch replaces sketches.Sketch (sketch.go:45-77).
type Sketch struct {
	codes []uint64
	pos   []uint32
	i     int
	idx   int
	rev   bool
}

func (r *Result) Sketch(i int) (*Sketch, error) {
	codes, pos, _, err := r.slice(i)
	if err != nil {
		return nil, err
	}
	return &Sketch{codes: codes, pos: pos, idx: -1}, nil
}
func (s *Sketch) Next() (uint64, bool) { // sketch.go:480
	if s.i >= len(s.codes) {
		return 0, false
	}
	c := s.codes[s.i]
	s.idx = int(s.pos[s.i] & 0x7fffffff)
	s.rev = s.pos[s.i]>>31 == 1
	s.i++
	return c, true
}
func (s *Sketch) NextMinimizer() (uint64, bool) { return s.Next() } // sketch.go:205
func (s *Sketch) NextSyncmer() (uint64, bool)   { return s.Next() } // sketch.go:312
func (s *Sketch) Index() int                    { return s.idx }    // sketch.go:488
// Strand is an extension: true iff the reverse-strand hash was the canonical one.
func (s *Sketch) Strand() bool { return s.rev }

// IdxValues: the tuples of record i as the reference's exported pair type (sketch.go:496).
func (r *Result) IdxValues(i int) ([]IdxValue, error) {
	codes, pos, _, err := r.slice(i)
	if err != nil {
		return nil, err
	}
	out := make([]IdxValue, len(codes))
	for j, c := range codes {
		p := j
		if pos != nil {
			p = int(pos[j] & 0x7fffffff)
		}
		out[j] = IdxValue{Idx: p, Val: c}
	}
	return out, nil
}

// ProteinIterator replaces sketches.ProteinIterator (iterator-protein.go:35): Next / Index only, as upstream.
type ProteinIterator struct {
	codes  []uint64
	i, idx int
}

func (r *Result) ProteinIterator(i int) (*ProteinIterator, error) {
	codes, _, _, err := r.slice(i)
	if err != nil {
		return nil, err
	}
	return &ProteinIterator{codes: codes, idx: -1}, nil
}
func (iter *ProteinIterator) Next() (code uint64, ok bool) { // iterator-protein.go:76
	if iter.i >= len(iter.codes) {
		return 0, false
	}
	code = iter.codes[iter.i]
	iter.idx = iter.i
	iter.i++
	return code, true
}
func (iter *ProteinIterator) Index() int { return iter.idx } // iterator-protein.go:93

// ProteinMinimizerSketch replaces sketches.ProteinMinimizerSketch (sketch-protein.go:32).
type ProteinMinimizerSketch struct {
	codes []uint64
	pos   []uint32
	i     int
	idx   int
}

func (r *Result) ProteinMinimizerSketch(i int) (*ProteinMinimizerSketch, error) {
	codes, pos, _, err := r.slice(i)
	if err != nil {
		return nil, err
	}
	return &ProteinMinimizerSketch{codes: codes, pos: pos, idx: -1}, nil
}
func (s *ProteinMinimizerSketch) Next() (code uint64, ok bool) { // sketch-protein.go:106
	if s.i >= len(s.codes) {
		return 0, false
	}
	code = s.codes[s.i]
	s.idx = int(s.pos[s.i] & 0x7fffffff)
	s.i++
	return code, true
}
func (s *ProteinMinimizerSketch) Index() int { return s.idx } // sketch-protein.go:213

// ---- the reference's single-sequence constructors: a batch of one on the default engine ----
var defaultEngine *Engine

func engine() (*Engine, error) {
	if defaultEngine == nil {
		e, err := NewEngine(0)
		if err != nil {
			return nil, err
		}
		defaultEngine = e
	}
	return defaultEngine, nil
}

func one(s *seq.Seq) (*Batch, error) {
	e, err := engine()
	if err != nil {
		return nil, err
	}
	return e.NewBatchFromSeqs([]*seq.Seq{s})
}

func NewHashIterator(s *seq.Seq, k int, canonical bool, circular bool) (*Iterator, error) { // iterator.go:615
	b, err := one(s)
	if err != nil {
		return nil, err
	}
	r, err := b.HashIterators(k, canonical, circular)
	if err != nil {
		return nil, err
	}
	return r.Iterator(0)
}

func NewKmerIterator(s *seq.Seq, k int, canonical bool, circular bool) (*Iterator, error) { // iterator.go:668
	b, err := one(s)
	if err != nil {
		return nil, err
	}
	r, err := b.KmerIterators(k, canonical, circular)
	if err != nil {
		return nil, err
	}
	it, err := r.Iterator(0)
	if err == nil && !canonical {
		it.perStr = len(it.codes) / 2
	}
	return it, err
}

func NewSimHashIterator(s *seq.Seq, k int, m int, scale int, canonical bool, circular bool) (*Iterator, error) { // iterator.go:113
	b, err := one(s)
	if err != nil {
		return nil, err
	}
	r, err := b.SimHashIterators(k, m, scale, canonical, circular)
	if err != nil {
		return nil, err
	}
	return r.Iterator(0)
}

func NewMinimizerSketch(S *seq.Seq, k int, w int, circular bool) (*Sketch, error) { // sketch.go:85
	b, err := one(S)
	if err != nil {
		return nil, err
	}
	r, err := b.MinimizerSketches(k, w, circular)
	if err != nil {
		return nil, err
	}
	return r.Sketch(0)
}

func NewSyncmerSketch(S *seq.Seq, k int, s int, circular bool) (*Sketch, error) { // sketch.go:142
	b, err := one(S)
	if err != nil {
		return nil, err
	}
	r, err := b.SyncmerSketches(k, s, circular)
	if err != nil {
		return nil, err
	}
	return r.Sketch(0)
}

func NewProteinIterator(s *seq.Seq, k int, codonTable int, frame int) (*ProteinIterator, error) { // iterator-protein.go:46
	b, err := one(s)
	if err != nil {
		return nil, err
	}
	r, err := b.ProteinIterators(k, codonTable, frame)
	if err != nil {
		return nil, err
	}
	return r.ProteinIterator(0)
}

func NewProteinMinimizerSketch(S *seq.Seq, k int, codonTable int, frame int, w int) (*ProteinMinimizerSketch, error) { // sketch-protein.go:62
	if k >= 1 && len(S.Seq) < k*3 { // upstream's order: k, then this length check (:66), only then w (:69)
		return nil, ErrShortSeq
	}
	b, err := one(S)
	if err != nil {
		return nil, err
	}
	r, err := b.ProteinMinimizerSketches(k, codonTable, frame, w)
	if err != nil {
		return nil, err
	}
	return r.ProteinMinimizerSketch(0)
}
