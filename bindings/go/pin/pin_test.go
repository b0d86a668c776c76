//go:build pin

// Pin harness: runs every case of tests/golden/sketches_golden.json through the REAL Go iterators of shenwei356/bio (no cgo, no
// device) and writes what they yield as JSON lines; scripts/pin_diff.py compares that with the committed goldens (the CPU oracle's
// values) and says which "parity unpinned" banners can go.  This image has no Go toolchain, so the file has never been compiled
// here; anyone with `go` pins the oracle in one command:
//
//	cd bindings/go/pin && go test -tags pin -run TestPin -v && python ../../../scripts/pin_diff.py pin_out.jsonl
//
// What it pins that the reference's own tests do not (SURVEY.md 8c): wyhash values (iterator-protein.go:87, sketch-protein.go:117),
// every syncmer value (sketch.go:312-477; TestSyncmer asserts nothing), the order of ties inside the first sorted window
// (sorts.Quicksort, sketch.go:236,351), ntHash of bytes outside ACGTacgt, k > 64.
package pin

import (
	"encoding/json"
	"errors"
	"fmt"
	"os"
	"testing"

	"github.com/shenwei356/bio/seq"
	"github.com/shenwei356/bio/sketches"
	"github.com/zeebo/wyhash"
)

type goldenCase struct {
	Name      string `json:"name"`
	Seq       string `json:"seq"`
	Fn        string `json:"fn"`
	K         int    `json:"k"`
	W         int    `json:"w"`
	S         int    `json:"s"`
	M         int    `json:"m"`
	Scale     int    `json:"scale"`
	Seed      uint64 `json:"seed"`
	Canonical *bool  `json:"canonical"`
	Circular  bool   `json:"circular"`
}

type goldenFile struct {
	Cases []goldenCase `json:"cases"`
}

type pinned struct {
	Name   string   `json:"name"`
	Fn     string   `json:"fn"`
	Values []uint64 `json:"values"`
	Index  []int    `json:"index,omitempty"`
	Error  string   `json:"error,omitempty"`
}

// the sentinel's name as the goldens spell it (include/biosketch.h bsk_err)
func errName(err error) string {
	switch {
	case err == nil:
		return ""
	case errors.Is(err, sketches.ErrInvalidK):
		return "ErrInvalidK"
	case errors.Is(err, sketches.ErrEmptySeq):
		return "ErrEmptySeq"
	case errors.Is(err, sketches.ErrShortSeq):
		return "ErrShortSeq"
	case errors.Is(err, sketches.ErrIllegalBase):
		return "ErrIllegalBase"
	case errors.Is(err, sketches.ErrKTooLarge):
		return "ErrKTooLarge"
	case errors.Is(err, sketches.ErrInvalidM):
		return "ErrInvalidM"
	case errors.Is(err, sketches.ErrInvalidScale):
		return "ErrInvalidScale"
	case errors.Is(err, sketches.ErrInvalidS):
		return "ErrInvalidS"
	case errors.Is(err, sketches.ErrInvalidW):
		return "ErrInvalidW"
	}
	return err.Error()
}

func canonical(c goldenCase) bool { return c.Canonical == nil || *c.Canonical }

// one case through the upstream iterator of its kind
func run(c goldenCase) pinned {
	out := pinned{Name: c.Name, Fn: c.Fn, Values: []uint64{}}
	alphabet := seq.DNAredundant
	if c.Fn == "protein_minimizer" || c.Fn == "protein_hashes" {
		alphabet = seq.Protein
	}
	// (no validation: the goldens hold N, lower case and other bytes on purpose)
	s, err := seq.NewSeqWithoutValidation(alphabet, []byte(c.Seq))
	if err != nil {
		out.Error = err.Error()
		return out
	}
	switch c.Fn {
	case "wyhash": // github.com/zeebo/wyhash v0.0.1, as called at iterator-protein.go:87
		out.Values = append(out.Values, wyhash.Hash([]byte(c.Seq), c.Seed))
	case "minimizer":
		sk, err := sketches.NewMinimizerSketch(s, c.K, c.W, c.Circular)
		if err != nil {
			out.Error = errName(err)
			return out
		}
		for {
			code, ok := sk.NextMinimizer()
			if !ok {
				break
			}
			out.Values = append(out.Values, code)
			out.Index = append(out.Index, sk.Index())
		}
	case "syncmer":
		sk, err := sketches.NewSyncmerSketch(s, c.K, c.S, c.Circular)
		if err != nil {
			out.Error = errName(err)
			return out
		}
		for {
			code, ok := sk.NextSyncmer()
			if !ok {
				break
			}
			out.Values = append(out.Values, code)
			out.Index = append(out.Index, sk.Index())
		}
	case "nthash":
		it, err := sketches.NewHashIterator(s, c.K, canonical(c), c.Circular)
		if err != nil {
			out.Error = errName(err)
			return out
		}
		for {
			code, ok := it.NextHash()
			if !ok {
				break
			}
			out.Values = append(out.Values, code)
		}
	case "kmer":
		it, err := sketches.NewKmerIterator(s, c.K, canonical(c), c.Circular)
		if err != nil {
			out.Error = errName(err)
			return out
		}
		for {
			code, ok, err := it.NextKmer()
			if err != nil {
				out.Error = errName(err)
				break
			}
			if !ok {
				break
			}
			out.Values = append(out.Values, code)
		}
	case "simhash":
		it, err := sketches.NewSimHashIterator(s, c.K, c.M, c.Scale, canonical(c), c.Circular)
		if err != nil {
			out.Error = errName(err)
			return out
		}
		for {
			code, ok := it.NextSimHash()
			if !ok {
				break
			}
			out.Values = append(out.Values, code)
		}
	case "protein_hashes":
		it, err := sketches.NewProteinIterator(s, c.K, 1, 1)
		if err != nil {
			out.Error = errName(err)
			return out
		}
		for {
			code, ok := it.Next()
			if !ok {
				break
			}
			out.Values = append(out.Values, code)
		}
	case "protein_minimizer":
		sk, err := sketches.NewProteinMinimizerSketch(s, c.K, 1, 1, c.W)
		if err != nil {
			out.Error = errName(err)
			return out
		}
		for {
			code, ok := sk.Next()
			if !ok {
				break
			}
			out.Values = append(out.Values, code)
			out.Index = append(out.Index, sk.Index())
		}
	default:
		out.Error = fmt.Sprintf("pin harness: unknown fn %q", c.Fn)
	}
	return out
}

func TestPin(t *testing.T) {
	golden := os.Getenv("BSK_GOLDEN")
	if golden == "" {
		golden = "../../../tests/golden/sketches_golden.json"
	}
	raw, err := os.ReadFile(golden)
	if err != nil {
		t.Fatalf("read %s: %v", golden, err)
	}
	var g goldenFile
	if err := json.Unmarshal(raw, &g); err != nil {
		t.Fatalf("parse %s: %v", golden, err)
	}
	dst := os.Getenv("BSK_PIN_OUT")
	if dst == "" {
		dst = "pin_out.jsonl"
	}
	f, err := os.Create(dst)
	if err != nil {
		t.Fatal(err)
	}
	defer f.Close()
	enc := json.NewEncoder(f)
	for _, c := range g.Cases {
		if err := enc.Encode(run(c)); err != nil {
			t.Fatal(err)
		}
	}
	t.Logf("%d cases -> %s", len(g.Cases), dst)
}
