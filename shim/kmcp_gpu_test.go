//go:build cgo && kmcpgpu

// Test of the cgo binding against the fixture of shim/testdata (made by tests/golden/make_shim_fixture.py of the kmcp-gpu
// repository: a 12-column database in two .uniki blocks, 48 reads, and the TSV `kmcp search` prints for them with default
// flags).  Run through shim/build.sh, which copies this file next to kmcp_gpu.go into kmcp/cmd/ and sets KMCP_GPU_FIXTURE.
// Both doors are exercised: one batch through SearchBatch, and the engine (NewGPUSearchEngine: sg.InCh -> sg.OutCh) the way
// search.go drives it.  NOTE: like kmcp_gpu.go this file could not be compiled where it was written (no Go toolchain).
package cmd

import (
	"bufio"
	"io"
	"os"
	"path/filepath"
	"sort"
	"strconv"
	"strings"
	"testing"

	"github.com/shenwei356/bio/seqio/fastx"
)

func gpuFixtureOptions() SearchOptions { // the defaults of `kmcp search` (search.go:1052-1102), single-end
	return SearchOptions{Threads: 4, DeduplicateThreshold: 256, SortBy: "qcov", MinQLen: 30, MinMatched: 10,
		MinQueryCov: 0.55, MinTargetCov: 0, MaxFPR: 0.01, FPRBufSize: 249}
}

func gpuFixtureQueries(t *testing.T, file string) []*Query {
	reader, err := fastx.NewDefaultReader(file)
	if err != nil {
		t.Fatal(err)
	}
	var queries []*Query
	for {
		record, err := reader.Read()
		if err == io.EOF {
			break
		}
		if err != nil {
			t.Fatal(err)
		}
		queries = append(queries, &Query{Idx: uint64(len(queries)), ID: append([]byte{}, record.ID...), Seq: record.Seq.Clone()})
	}
	return queries
}

// rows exactly as search.go:517-575 writes them
func gpuFixtureRows(results []*QueryResult) (rows []string, matched int) {
	for _, r := range results {
		if r.Matches == nil {
			continue
		}
		matched++
		for _, m := range *r.Matches {
			rows = append(rows, strings.Join([]string{
				string(r.QueryID), strconv.Itoa(r.QueryLen), strconv.Itoa(r.NumKmers), strconv.FormatFloat(m.FPR, 'e', 4, 64),
				strconv.Itoa(len(*r.Matches)), m.Target[0], strconv.Itoa(int(uint16(m.TargetIdx[0]))), strconv.Itoa(int(m.TargetIdx[0] >> 16)),
				strconv.Itoa(int(m.GenomeSize[0])), strconv.Itoa(r.K), strconv.Itoa(m.NumKmers),
				strconv.FormatFloat(m.QCov, 'f', 4, 64), strconv.FormatFloat(m.TCov, 'f', 4, 64), strconv.FormatFloat(m.JaccardIndex, 'f', 4, 64),
				strconv.Itoa(int(r.QueryIdx))}, "\t"))
		}
	}
	return
}

func gpuFixtureExpected(t *testing.T, file string) (rows []string, matched int) {
	fh, err := os.Open(file)
	if err != nil {
		t.Fatal(err)
	}
	defer fh.Close()
	sc := bufio.NewScanner(fh)
	for sc.Scan() {
		line := sc.Text()
		if strings.HasPrefix(line, "# matched queries: ") {
			matched, _ = strconv.Atoi(strings.TrimPrefix(line, "# matched queries: "))
		}
		if line == "" || line[0] == '#' {
			continue
		}
		rows = append(rows, line)
	}
	return
}

func gpuFixtureCompare(t *testing.T, got, want []string, gotMatched, wantMatched int) {
	if gotMatched != wantMatched {
		t.Errorf("matched queries: got %d, want %d", gotMatched, wantMatched)
	}
	if len(got) != len(want) {
		t.Fatalf("rows: got %d, want %d", len(got), len(want))
	}
	for i := range want {
		if got[i] != want[i] {
			t.Errorf("row %d:\n got  %s\n want %s", i, got[i], want[i])
		}
	}
}

func TestGPUSearchFixture(t *testing.T) {
	dir := os.Getenv("KMCP_GPU_FIXTURE")
	if dir == "" {
		t.Skip("KMCP_GPU_FIXTURE is not set (shim/build.sh sets it to <kmcp-gpu repository>/shim/testdata)")
	}
	want, wantMatched := gpuFixtureExpected(t, filepath.Join(dir, "expected.tsv"))
	dbPath := filepath.Join(dir, "db", "R001")

	// 1. one batch through Submit + Wait
	db, err := OpenGPUDB(dbPath, 0)
	if err != nil {
		t.Fatal(err)
	}
	queries := gpuFixtureQueries(t, filepath.Join(dir, "reads.fq"))
	results, err := db.SearchBatch(queries, gpuFixtureOptions(), 0)
	if err != nil {
		t.Fatal(err)
	}
	got, matched := gpuFixtureRows(results)
	gpuFixtureCompare(t, got, want, matched, wantMatched)
	if err = db.Close(); err != nil {
		t.Fatal(err)
	}

	// 2. the engine as search.go drives it: queries into sg.InCh, results out of sg.OutCh in any order
	sg, gdb, err := NewGPUSearchEngine(gpuFixtureOptions(), 0, dbPath)
	if err != nil {
		t.Fatal(err)
	}
	if len(sg.DBs) != 1 || sg.DBs[0].Info.K != 21 {
		t.Fatalf("engine stub: %d database(s), k = %d", len(sg.DBs), sg.DBs[0].Info.K)
	}
	var all []*QueryResult
	collected := make(chan int)
	go func() {
		for r := range sg.OutCh {
			all = append(all, r)
		}
		collected <- 1
	}()
	for _, q := range gpuFixtureQueries(t, filepath.Join(dir, "reads.fq")) {
		pq := poolQuery.Get().(*Query) // RunGPUEngine recycles queries and sequences into the engine's pools
		pq.Idx, pq.ID, pq.Seq, pq.Seq2 = q.Idx, q.ID, q.Seq, nil
		sg.InCh <- pq
	}
	close(sg.InCh)
	sg.Wait()
	<-collected
	sort.Slice(all, func(i, j int) bool { return all[i].QueryIdx < all[j].QueryIdx })
	got, matched = gpuFixtureRows(all)
	gpuFixtureCompare(t, got, want, matched, wantMatched)
	if err = sg.Close(); err != nil {
		t.Fatal(err)
	}
	if err = gdb.Close(); err != nil {
		t.Fatal(err)
	}
}
