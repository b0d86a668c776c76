// The shim REPLACES the reference package (INTEGRATION.md 2): an application adds
//   replace github.com/shenwei356/bio/sketches => <this directory>
module github.com/shenwei356/bio/sketches

go 1.17

require github.com/shenwei356/bio v0.13.8
