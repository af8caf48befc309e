//go:build cgo && kmcpgpu

// Package cmd — cgo binding of libkmcpgpu.so for `kmcp search` (build with: CGO_ENABLED=1 go build -tags kmcpgpu).
//
// Drop this file into kmcp/cmd/ of shenwei356/kmcp v0.9.5.  It replaces the per-query fan-out of
// UnikIndexDB.handleQuery (util-db-search.go:763-1025) by batched calls into the MI355X library while keeping
// the engine's channel protocol (sg.InCh chan *Query -> sg.OutCh chan *QueryResult, util-db-search.go:201-202),
// so search.go (reader loop :793-1000, ordered writer :448-588, :733-781) is untouched.
//
// NOTE: there is no Go toolchain in the build container of this repository, so this file is delivered as source
// and has not been compiled there.  The exact sequence of C calls it makes — malloc'd CSR buffers, kmcpg_submit,
// kmcpg_wait_pairs from other threads, kmcpg_expand_pairs per matched query into a malloc'd scratch array,
// kmcpg_result_pairs_free, the error fetch on the failing thread — is replayed from several threads by tests/shim_replay.c
// (tests/test_gpu_shim_replay.py), which is compiled and run with every GPU test run.
package cmd

// Tree layout the #cgo lines assume (shim/build.sh puts the files there): this file and kmcp_gpu_test.go in
// <kmcp source>/kmcp/cmd/, include/kmcp_gpu.h of this repository in <kmcp source>/include/; the library is found through
// CGO_LDFLAGS="-L<this repository>/kmcp_amd -Wl,-rpath,<this repository>/kmcp_amd".

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -lkmcpgpu
#include <stdlib.h>
#include "kmcp_gpu.h"
*/
import "C"

import (
	"fmt"
	"path/filepath"
	"runtime"
	"sync"
	"unsafe"
)

// GPUBatchSize is the number of queries handed to the GPU per call.
const GPUBatchSize = 1 << 17

// GPUInFlight bounds the batches between kmcpg_submit and kmcpg_wait (the library has KMCPG_INFLIGHT = 4 lanes; the
// reference bounds its in-flight queries the same way with a token channel, util-db-search.go:243, :347-351).
const GPUInFlight = 3

// GPUDB wraps a kmcpg_db handle (one per database directory, i.e. per <db>/R001).
type GPUDB struct {
	h     *C.kmcpg_db
	Info  C.kmcpg_info
	names []string // Header.Names of every column, fetched once: no cgo call per match later
}

// gpuCall runs one C call and, if it fails, fetches the library's thread-local message on the SAME OS thread (the Go
// scheduler may otherwise move the goroutine between the two cgo calls).
func gpuCall(f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := f(); rc != 0 {
		return fmt.Errorf("kmcp-gpu: %s (code %d)", C.GoString(C.kmcpg_last_error()), int(rc))
	}
	return nil
}

func (db *GPUDB) finishOpen() error {
	if err := gpuCall(func() C.int { return C.kmcpg_db_info(db.h, &db.Info) }); err != nil {
		return err
	}
	db.names = make([]string, int(db.Info.n_cols))
	for c := range db.names {
		var name *C.char
		if err := gpuCall(func() C.int { return C.kmcpg_col_info(db.h, C.uint32_t(c), &name, nil, nil, nil) }); err != nil {
			return err
		}
		db.names[c] = C.GoString(name)
	}
	return nil
}

// OpenGPUDB replaces NewUnikIndexDB (util-db-search.go:648-743): parses __db.yml and the .uniki headers and
// makes the bit matrices resident in HBM.
func OpenGPUDB(path string, device int) (*GPUDB, error) {
	cpath := C.CString(path)
	defer C.free(unsafe.Pointer(cpath))
	opts := C.kmcpg_opts{device: C.int32_t(device), shard_rank: 0, shard_count: 1}
	db := &GPUDB{}
	if err := gpuCall(func() C.int { return C.kmcpg_open(cpath, &opts, &db.h) }); err != nil {
		return nil, err
	}
	if err := db.finishOpen(); err != nil {
		db.Close()
		return nil, err
	}
	return db, nil
}

// OpenGPUDBDevices is OpenGPUDB over several GPUs of the node: the index blocks are partitioned over `devices` and every
// batch fans out to all of them (one host thread per GPU inside the library).
func OpenGPUDBDevices(path string, devices []int32) (*GPUDB, error) {
	cpath := C.CString(path)
	defer C.free(unsafe.Pointer(cpath))
	cdev := (*C.int32_t)(C.malloc(C.size_t(4 * len(devices)))) // no Go pointer handed to C
	defer C.free(unsafe.Pointer(cdev))
	copy(unsafe.Slice((*int32)(unsafe.Pointer(cdev)), len(devices)), devices)
	db := &GPUDB{}
	if err := gpuCall(func() C.int { return C.kmcpg_open_devices(cpath, cdev, C.int32_t(len(devices)), &db.h) }); err != nil {
		return nil, err
	}
	if err := db.finishOpen(); err != nil {
		db.Close()
		return nil, err
	}
	return db, nil
}

// OpenGPUDBPaged is OpenGPUDB for an index larger than the GPU's memory: the counterpart of the reference's mmap /
// --low-mem modes (util-db-search.go:1238-1280, search.go:80).  The index is searched one resident part after the other
// (passes = 0: as few parts as fit); batches should then be large, RunGPUEngine's batchSize is raised by the caller.
// OpenGPUDB reports an index that does not fit with an error that names the bytes needed and free (code -5).
func OpenGPUDBPaged(path string, device int, passes int) (*GPUDB, error) {
	cpath := C.CString(path)
	defer C.free(unsafe.Pointer(cpath))
	db := &GPUDB{}
	if err := gpuCall(func() C.int { return C.kmcpg_open_paged(cpath, C.int32_t(device), C.int32_t(passes), &db.h) }); err != nil {
		return nil, err
	}
	if err := db.finishOpen(); err != nil {
		db.Close()
		return nil, err
	}
	return db, nil
}

// ExchangeInfo says how a multi-GPU handle brings the shards' hit lists together ("RCCL gather over N device(s)" or
// "host merge ... (reason)").
func (db *GPUDB) ExchangeInfo() string { return C.GoString(C.kmcpg_exchange_info(db.h)) }

// Close replaces UnikIndexDB.Close (util-db-search.go:1119-1150).
func (db *GPUDB) Close() error { return gpuCall(func() C.int { return C.kmcpg_close(db.h) }) }

func gpuParams(opt SearchOptions) C.kmcpg_params {
	sortBy := 0
	switch opt.SortBy {
	case "tcov":
		sortBy = 1
	case "jacc":
		sortBy = 2
	}
	b2i := func(b bool) C.int32_t {
		if b {
			return 1
		}
		return 0
	}
	return C.kmcpg_params{
		min_qlen: C.int32_t(opt.MinQLen), min_matched: C.int32_t(opt.MinMatched),
		min_qcov: C.double(opt.MinQueryCov), min_tcov: C.double(opt.MinTargetCov), max_fpr: C.double(opt.MaxFPR),
		dedup_threshold: C.int32_t(opt.DeduplicateThreshold), try_se: b2i(opt.TrySingleEnd),
		sort_by: C.int32_t(sortBy), do_not_sort: b2i(opt.DoNotSort), top_n_scores: C.int32_t(opt.TopNScores),
		fpr_buf_size: C.int32_t(opt.FPRBufSize),
		// k = 0: the library walks the database's k-mer sizes, largest first, as handleQuery does (:764, :1016-1022)
	}
}

// gpuBatch is one batch between Submit and Wait.
type gpuBatch struct {
	queries []*Query
	ticket  *C.kmcpg_ticket
}

// Submit packs the pooled Querys into two CSR buffers in C memory (no Go pointer is retained by C), hands them to
// kmcpg_submit — which copies them into pinned staging and enqueues the GPU work — and frees them again.
// kmcpg_submit never blocks; it fails with KMCPG_EBUSY when all lanes are taken, which RunGPUEngine's tokens rule out.
func (db *GPUDB) Submit(queries []*Query, opt SearchOptions) (*gpuBatch, error) {
	n := len(queries)
	paired := n > 0 && queries[0].Seq2 != nil
	pack := func(get func(q *Query) []byte) (unsafe.Pointer, unsafe.Pointer, func()) {
		total := 0
		for _, q := range queries {
			total += len(get(q))
		}
		seqs := C.malloc(C.size_t(total + 1))
		offs := C.malloc(C.size_t(8 * (n + 1)))
		sb := unsafe.Slice((*byte)(seqs), total+1)
		ob := unsafe.Slice((*uint64)(offs), n+1)
		p := 0
		for i, q := range queries {
			ob[i] = uint64(p)
			p += copy(sb[p:], get(q))
		}
		ob[n] = uint64(p)
		return seqs, offs, func() { C.free(seqs); C.free(offs) }
	}
	s1, o1, free1 := pack(func(q *Query) []byte { return q.Seq.Seq })
	defer free1()
	var s2, o2 unsafe.Pointer
	if paired {
		var free2 func()
		s2, o2, free2 = pack(func(q *Query) []byte { return q.Seq2.Seq })
		defer free2()
	}
	params := gpuParams(opt)
	b := &gpuBatch{queries: queries}
	err := gpuCall(func() C.int {
		return C.kmcpg_submit(db.h, (*C.uint8_t)(s1), (*C.uint64_t)(o1), (*C.uint8_t)(s2), (*C.uint64_t)(o2), C.uint32_t(n), &params, &b.ticket)
	})
	if err != nil {
		return nil, err
	}
	return b, nil
}

// Wait blocks until the batch's GPU work is done and takes the batch's FINAL matches as compact (column, mKmers) pairs
// (kmcpg_wait_pairs, round 5: every threshold, -f and --keep-top-scores applied, in print order; the --try-se / smaller-k
// retries run inside the call on this thread).  The float64 values of a Match — qCov, tCov, jacc (util-db-search.go:7487-7489)
// and the FPR column (util-fpr.go:32-50) — are derived one query at a time by kmcpg_expand_pairs into a small scratch array and
// copied into the engine's own QueryResult/Match structs (util-db-search.go:60-93): the 56-byte records of a batch (1.5 GB per
// 131 072 reads on a database full of close relatives) never exist as a whole.  Everything C returns is copied before
// kmcpg_result_pairs_free.  The call sequence is replayed by tests/shim_replay.c (three threads, ASan) with every GPU test run.
func (db *GPUDB) Wait(b *gpuBatch, dbID int) ([]*QueryResult, error) {
	var res C.kmcpg_result_pairs
	if err := gpuCall(func() C.int { return C.kmcpg_wait_pairs(b.ticket, &res) }); err != nil { // the ticket is consumed either way
		return nil, err
	}
	defer C.kmcpg_result_pairs_free(&res)
	n := len(b.queries)
	out := make([]*QueryResult, n)
	if n == 0 {
		return out, nil
	}
	qlen := unsafe.Slice((*int32)(unsafe.Pointer(res.qlen)), n)
	qk := unsafe.Slice((*int32)(unsafe.Pointer(res.qkmers)), n)
	ks := unsafe.Slice((*int32)(unsafe.Pointer(res.ksize)), n)
	offs := unsafe.Slice((*uint64)(unsafe.Pointer(res.match_offs)), n+1)
	var pairs []C.kmcpg_pair
	if offs[n] > 0 {
		pairs = unsafe.Slice(res.pairs, int(offs[n]))
	}
	// one query's records at a time, in C memory (no Go pointer crosses the boundary; grown to the largest query of the batch)
	scratchCap := 256
	scratch := (*C.kmcpg_match)(C.malloc(C.size_t(scratchCap) * C.size_t(unsafe.Sizeof(C.kmcpg_match{}))))
	defer func() { C.free(unsafe.Pointer(scratch)) }()
	for i, q := range b.queries {
		r := poolQueryResult.Get().(*QueryResult)
		r.QueryIdx, r.QueryID, r.QueryLen = q.Idx, q.ID, int(qlen[i])
		r.DBId, r.K, r.NumKmers, r.Matches = dbID, int(ks[i]), int(qk[i]), nil
		if m := int(offs[i+1] - offs[i]); m > 0 {
			if m > scratchCap {
				C.free(unsafe.Pointer(scratch))
				scratchCap = m + m/2
				scratch = (*C.kmcpg_match)(C.malloc(C.size_t(scratchCap) * C.size_t(unsafe.Sizeof(C.kmcpg_match{}))))
			}
			first := &pairs[offs[i]]
			if err := gpuCall(func() C.int { return C.kmcpg_expand_pairs(db.h, C.int32_t(qk[i]), first, C.uint64_t(m), scratch) }); err != nil {
				return nil, err
			}
			matches := poolMatches.Get().(*[]*Match)
			// a pooled slice keeps whatever length its last user left it with unless every consumer resets it before Put (the
			// reference does, search.go:582-583): reset here as well, so that the shim does not depend on it
			*matches = (*matches)[:0]
			for _, mm := range unsafe.Slice(scratch, m) {
				*matches = append(*matches, &Match{
					Target: []string{db.names[int(mm.col)]}, TargetIdx: []uint32{uint32(mm.target_idx)},
					GenomeSize: []uint64{uint64(mm.gsize)}, NumKmers: int(mm.mkmers), FPR: float64(mm.fpr),
					QCov: float64(mm.qcov), TCov: float64(mm.tcov), JaccardIndex: float64(mm.jacc),
				})
			}
			r.Matches = matches
		}
		out[i] = r
	}
	return out, nil
}

// SearchBatch = Submit + Wait (one batch, nothing overlapped).
func (db *GPUDB) SearchBatch(queries []*Query, opt SearchOptions, dbID int) ([]*QueryResult, error) {
	b, err := db.Submit(queries, opt)
	if err != nil {
		return nil, err
	}
	return db.Wait(b, dbID)
}

// RunGPUEngine is the batcher that takes the place of the goroutine started in
// NewUnikIndexDBSearchEngine (util-db-search.go:239-352).  One goroutine drains sg.InCh into batches and submits them
// (packing batch i+1 while the GPU works on batch i); two goroutines wait for tickets, convert the results and re-emit
// the QueryResults into sg.OutCh (conversion of batch i while batch i+1 is on the GPU).  search.go's reorder goroutine
// (:733-781) restores input order, so batches may finish in any order.
func RunGPUEngine(sg *UnikIndexDBSearchEngine, db *GPUDB) {
	inflight := make(chan *gpuBatch, GPUInFlight)
	tokens := make(chan struct{}, GPUInFlight) // one per batch between Submit and the end of Wait: never more than the library's lanes
	var wg sync.WaitGroup
	for w := 0; w < 2; w++ {
		wg.Add(1)
		go func() {
			defer wg.Done()
			for b := range inflight {
				results, err := db.Wait(b, 0)
				<-tokens
				checkError(err) // fatal, as every error on the reference's search path (util-cli.go:35-40)
				for i, r := range results {
					if r.Matches != nil && len(sg.Options.NameMap) > 0 { // name mapping, util-db-search.go:317-332
						for _, m := range *r.Matches {
							if t, ok := sg.Options.NameMap[m.Target[0]]; ok {
								m.Target[0] = t
							}
						}
					}
					sg.OutCh <- r
					q := b.queries[i]
					poolSeq.Put(q.Seq)
					if q.Seq2 != nil {
						poolSeq.Put(q.Seq2)
					}
					poolQuery.Put(q)
				}
			}
		}()
	}
	go func() {
		batch := make([]*Query, 0, GPUBatchSize)
		flush := func() {
			tokens <- struct{}{} // blocks while GPUInFlight batches are in flight: kmcpg_submit never sees all lanes taken
			b, err := db.Submit(batch, sg.Options)
			checkError(err)
			inflight <- b
			batch = make([]*Query, 0, GPUBatchSize)
		}
		for q := range sg.InCh {
			batch = append(batch, q)
			if len(batch) == GPUBatchSize {
				flush()
			}
		}
		if len(batch) > 0 {
			flush()
		}
		close(inflight)
		wg.Wait()
		sg.done <- 1
	}()
}

// NewGPUSearchEngine takes the place of NewUnikIndexDBSearchEngine (util-db-search.go:222-584) for one database that is searched
// on the GPU: search.go keeps talking to sg.InCh / sg.OutCh / sg.Wait() / sg.Close() and reading sg.DBs[0].Info (the
// min-query-cov check :405-409, the k-mer sizes :790).  The UnikIndexDB it finds there is a stub that carries the parsed
// __db.yml only: no index file is opened or mapped by Go, the matrices live in HBM behind `gdb`.
// Wiring in search.go:400:
//
//	var sg *UnikIndexDBSearchEngine
//	if os.Getenv("KMCP_GPU") != "" {
//		var gdb *GPUDB
//		sg, gdb, err = NewGPUSearchEngine(searchOpt, 0, dbDirs[0])
//		if err == nil { defer gdb.Close() }
//	} else {
//		sg, err = NewUnikIndexDBSearchEngine(searchOpt, dbDirs...)
//	}
func NewGPUSearchEngine(opt SearchOptions, device int, dbPath string) (*UnikIndexDBSearchEngine, *GPUDB, error) {
	info, err := UnikIndexDBInfoFromFile(filepath.Join(dbPath, dbInfoFile))
	if err != nil {
		return nil, nil, err
	}
	if err = info.Check(); err != nil {
		return nil, nil, err
	}
	gdb, err := OpenGPUDB(dbPath, device)
	if err != nil {
		return nil, nil, err
	}
	// UnikIndexDB.Close (:1119-1150) closes InCh, waits for `done` and closes the (here: no) index files
	stub := &UnikIndexDB{Options: opt, path: dbPath, Info: info, InCh: make(chan *Query), done: make(chan int, 1)}
	stub.done <- 1
	sg := &UnikIndexDBSearchEngine{Options: opt, DBs: []*UnikIndexDB{stub}, DBNames: []string{filepath.Base(dbPath)}}
	sg.done = make(chan int)
	sg.InCh = make(chan *Query, 2*GPUBatchSize)
	sg.OutCh = make(chan *QueryResult, 2*GPUBatchSize)
	RunGPUEngine(sg, gdb)
	return sg, gdb, nil
}
