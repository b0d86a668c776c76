#!/bin/bash
# One command, on ANY box with network access (the build image of this repository has none and no Go): run every case of
# tests/golden/sketches_golden.json through the REAL shenwei356/bio iterators and say which "parity unpinned" banners can go.
#
#     bindings/go/pin/run.sh            # needs `go` >= 1.22 on PATH, or docker / podman (then it uses the golang:1.22 image)
#
# It writes bindings/go/pin/pin_out.jsonl (the upstream values), prints scripts/pin_diff.py's verdict banner by banner, and -- with
# --apply -- lets scripts/pin_apply.py rewrite the banners in oracle/bio_oracle.h, include/biosketch.h and DESIGN.md for the banners that
# were confirmed (nothing is touched for a banner with a differing case: pin_diff prints the cases instead).
# The module proxy is the only thing fetched: github.com/shenwei356/bio v0.13.8 and its go.sum-pinned dependencies (will-rowe/nthash
# v0.4.0, zeebo/wyhash v0.0.1, twotwotwo/sorts, shenwei356/kmers v0.1.0).  No cgo, no GPU, nothing of this repository is compiled.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../../.." && pwd)"
APPLY=0
[ "${1:-}" = "--apply" ] && APPLY=1
run_go() {
  (cd "$HERE" && go mod tidy && go test -tags pin -run TestPin -count=1 -v)
}
if command -v go >/dev/null 2>&1; then
  run_go
elif command -v docker >/dev/null 2>&1 || command -v podman >/dev/null 2>&1; then
  OCI=$(command -v docker || command -v podman)
  "$OCI" run --rm -v "$ROOT":/src -w /src/bindings/go/pin golang:1.22 bash -c "go mod tidy && go test -tags pin -run TestPin -count=1 -v"
else
  echo "neither go nor docker/podman found: install Go >= 1.22 (https://go.dev/dl) and re-run" >&2
  exit 2
fi
test -s "$HERE/pin_out.jsonl" || { echo "the harness wrote no pin_out.jsonl" >&2; exit 3; }
set +e
python3 "$ROOT/scripts/pin_diff.py" "$HERE/pin_out.jsonl" --json "$HERE/pin_verdict.json"
RC=$?
set -e
if [ "$APPLY" = 1 ]; then python3 "$ROOT/scripts/pin_apply.py" "$HERE/pin_verdict.json"; fi
exit $RC
