//go:build cgo && kmcpgpu

// Package cmd — cgo binding of libkmcpgpu.so for `kmcp search` (build with: CGO_ENABLED=1 go build -tags kmcpgpu).
//
// Drop this file into kmcp/cmd/ of shenwei356/kmcp v0.9.5.  It replaces the per-query fan-out of
// UnikIndexDB.handleQuery (util-db-search.go:763-1025) by batched calls into the MI355X library while keeping
// the engine's channel protocol (sg.InCh chan *Query -> sg.OutCh chan *QueryResult, util-db-search.go:201-202),
// so search.go (reader loop :793-1000, ordered writer :448-588, :733-781) is untouched.
//
// NOTE: there is no Go toolchain in the build container of this repository, so this file is delivered as source
// and has not been compiled there; the C ABI it binds (include/kmcp_gpu.h) is exercised by the C++ CLI and the
// Python tests.
package cmd

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -lkmcpgpu
#include <stdlib.h>
#include "kmcp_gpu.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// GPUBatchSize is the number of queries handed to the GPU per call.
const GPUBatchSize = 1 << 17

// GPUDB wraps a kmcpg_db handle (one per database directory, i.e. per <db>/R001).
type GPUDB struct {
	h    *C.kmcpg_db
	Info C.kmcpg_info
}

func gpuErr(rc C.int) error {
	if rc == 0 {
		return nil
	}
	return fmt.Errorf("kmcp-gpu: %s (code %d)", C.GoString(C.kmcpg_last_error()), int(rc))
}

// OpenGPUDB replaces NewUnikIndexDB (util-db-search.go:648-743): parses __db.yml and the .uniki headers and
// makes the bit matrices resident in HBM.
func OpenGPUDB(path string, device int) (*GPUDB, error) {
	cpath := C.CString(path)
	defer C.free(unsafe.Pointer(cpath))
	opts := C.kmcpg_opts{device: C.int32_t(device), shard_rank: 0, shard_count: 1}
	db := &GPUDB{}
	if err := gpuErr(C.kmcpg_open(cpath, &opts, &db.h)); err != nil {
		return nil, err
	}
	if err := gpuErr(C.kmcpg_db_info(db.h, &db.Info)); err != nil {
		return nil, err
	}
	return db, nil
}

// OpenGPUDBDevices is OpenGPUDB over several GPUs of the node: the index blocks are partitioned over `devices` and every
// SearchBatch call fans out to all of them (one host thread per GPU inside the library).
func OpenGPUDBDevices(path string, devices []int32) (*GPUDB, error) {
	cpath := C.CString(path)
	defer C.free(unsafe.Pointer(cpath))
	db := &GPUDB{}
	if err := gpuErr(C.kmcpg_open_devices(cpath, (*C.int32_t)(unsafe.Pointer(&devices[0])), C.int32_t(len(devices)), &db.h)); err != nil {
		return nil, err
	}
	if err := gpuErr(C.kmcpg_db_info(db.h, &db.Info)); err != nil {
		return nil, err
	}
	return db, nil
}

// Close replaces UnikIndexDB.Close (util-db-search.go:1119-1150).
func (db *GPUDB) Close() error { return gpuErr(C.kmcpg_close(db.h)) }

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
	}
}

// SearchBatch runs handleQuery's work for a batch of queries and converts the CSR result into the
// engine's own QueryResult/Match structs (util-db-search.go:60-93).  Everything C returns is copied
// before kmcpg_result_free, so no C pointer outlives the call and no Go pointer is retained by C.
func (db *GPUDB) SearchBatch(queries []*Query, opt SearchOptions, dbID int) ([]*QueryResult, error) {
	n := len(queries)
	if n == 0 {
		return nil, nil
	}
	paired := queries[0].Seq2 != nil
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
	var res C.kmcpg_result
	rc := C.kmcpg_search_batch(db.h, (*C.uint8_t)(s1), (*C.uint64_t)(o1), (*C.uint8_t)(s2), (*C.uint64_t)(o2),
		C.uint32_t(n), &params, &res)
	if err := gpuErr(rc); err != nil {
		return nil, err
	}
	defer C.kmcpg_result_free(&res)

	qlen := unsafe.Slice((*int32)(unsafe.Pointer(res.qlen)), n)
	qk := unsafe.Slice((*int32)(unsafe.Pointer(res.qkmers)), n)
	offs := unsafe.Slice((*uint64)(unsafe.Pointer(res.match_offs)), n+1)
	var ms []C.kmcpg_match
	if offs[n] > 0 {
		ms = unsafe.Slice(res.matches, int(offs[n]))
	}
	out := make([]*QueryResult, n)
	for i, q := range queries {
		r := poolQueryResult.Get().(*QueryResult)
		r.QueryIdx, r.QueryID, r.QueryLen = q.Idx, q.ID, int(qlen[i])
		r.DBId, r.K, r.NumKmers, r.Matches = dbID, int(res.k), int(qk[i]), nil
		if offs[i+1] > offs[i] {
			matches := poolMatches.Get().(*[]*Match)
			for _, m := range ms[offs[i]:offs[i+1]] {
				var name *C.char
				C.kmcpg_col_info(db.h, m.col, &name, nil, nil, nil)
				*matches = append(*matches, &Match{
					Target: []string{C.GoString(name)}, TargetIdx: []uint32{uint32(m.target_idx)},
					GenomeSize: []uint64{uint64(m.gsize)}, NumKmers: int(m.mkmers), FPR: float64(m.fpr),
					QCov: float64(m.qcov), TCov: float64(m.tcov), JaccardIndex: float64(m.jacc),
				})
			}
			r.Matches = matches
		}
		out[i] = r
	}
	return out, nil
}

// RunGPUEngine is the batcher that takes the place of the goroutine started in
// NewUnikIndexDBSearchEngine (util-db-search.go:239-352): it drains sg.InCh into batches, searches them on the
// GPU and re-emits the QueryResults into sg.OutCh.  search.go's reorder goroutine (:733-781) restores input order.
func RunGPUEngine(sg *UnikIndexDBSearchEngine, db *GPUDB) {
	go func() {
		batch := make([]*Query, 0, GPUBatchSize)
		flush := func() {
			results, err := db.SearchBatch(batch, sg.Options, 0)
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
				q := batch[i]
				poolSeq.Put(q.Seq)
				if q.Seq2 != nil {
					poolSeq.Put(q.Seq2)
				}
				poolQuery.Put(q)
			}
			batch = batch[:0]
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
		sg.done <- 1
	}()
}
