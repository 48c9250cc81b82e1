/*
 Licensed to the Apache Software Foundation (ASF) under one
 or more contributor license agreements.  See the NOTICE file
 distributed with this work for additional information
 regarding copyright ownership.  The ASF licenses this file
 to you under the Apache License, Version 2.0 (the
 "License"); you may not use this file except in compliance
 with the License.  You may obtain a copy of the License at

     http://www.apache.org/licenses/LICENSE-2.0

 Unless required by applicable law or agreed to in writing, software
 distributed under the License is distributed on an "AS IS" BASIS,
 WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND, either express or implied.
 See the License for the specific language governing permissions and
 limitations under the License.
*/

// Package predicates: gpu_predicate_manager.go is the drop-in for pkg/plugin/predicates of apache/yunikorn-k8shim.
//
// It implements the PredicateManager interface (predicate_manager.go:47-55) on top of the MI355X engine. The Go side is
// deliberately thin: pods and nodes cross the cgo boundary as the JSON the API machinery already produces for them
// (encoding/json on v1.Pod / v1.Node emits exactly the Kubernetes field names libykhost parses), libykhost.so keeps the
// encoded mirror of pkg/cache/external.SchedulerCache and drives libykpred.so (include/ykhost.h, include/ykpred.h).
// What stays in Go: the interface, the hooks in the cache's critical sections, the routing of asks the engine does not
// evaluate to the existing CPU manager, the configuration key and the counters.
//
// Place this file in pkg/plugin/predicates/, the two headers and libraries under third_party/ykpred/, and apply the three
// patches of integration/patches/ (context.go:130, the Observer hooks of scheduler_cache.go, the schedulerconf.go key):
// `git apply integration/patches/*.diff` from the root of the yunikorn-k8shim checkout.
// This file cannot be compiled in the build image of this repository (no Go toolchain); scripts/check_go_bindings.py
// checks every C symbol and constant it uses against the headers.
package predicates

/*
#cgo CFLAGS: -I${SRCDIR}/../../../third_party/ykpred/include
#cgo LDFLAGS: -L${SRCDIR}/../../../third_party/ykpred/lib -lykhost -lykpred -Wl,-rpath,${SRCDIR}/../../../third_party/ykpred/lib
#include <stdlib.h>
#include "ykhost.h"
#include "ykpred.h"
*/
import "C"

import (
	"encoding/json"
	"errors"
	"fmt"
	"strings"
	"sync"
	"sync/atomic"
	"unsafe"

	"go.uber.org/zap"
	v1 "k8s.io/api/core/v1"
	fwk "k8s.io/kube-scheduler/framework"
	"k8s.io/kubernetes/pkg/scheduler/framework"

	"github.com/apache/yunikorn-k8shim/pkg/log"
)

// Values of the configuration key service.predicateEngine (pkg/conf/schedulerconf.go, next to the other service.* keys).
const (
	PredicateEngineCPU = "cpu" // the reference's predicateManagerImpl (default)
	PredicateEngineGPU = "gpu" // this file; falls back to "cpu" when no device can be opened
)

// GPUPredicateCounters are exported through the scheduler's metrics / state dump.
type GPUPredicateCounters struct {
	DeviceAnswers     atomic.Int64 // Predicates() calls answered by the engine
	RoutedUnsupported atomic.Int64 // asks the engine does not evaluate (volumes, DRA claims, dictionary limits): CPU manager
	RoutedNotMirrored atomic.Int64 // pod or node not (yet) in the mirror: CPU manager
	RoutedOnError     atomic.Int64 // the engine reported an error: CPU manager
	PreemptionAnswers atomic.Int64
	MirrorUpdates     atomic.Int64
	MirrorErrors      atomic.Int64
}

// CacheObserver is what SchedulerCache calls from inside its own critical sections (six call sites, patch
// integration/patches/scheduler_cache.go.diff: scheduler_cache.go:149,190,304,392,444,464). The cache declares the same
// method set as external.Observer — pkg/plugin/support imports pkg/cache/external, so the interface the cache stores cannot
// live in this package; Go's structural typing makes *gpuPredicateManager satisfy both. The cache write lock is held, so
// mirror updates are serialised against each other exactly like the cache's; evaluations take the handle's own lock inside
// libykhost.
type CacheObserver interface {
	OnUpdateNode(node *v1.Node)
	OnRemoveNode(node *v1.Node)
	OnUpdatePod(pod *v1.Pod)
	OnRemovePod(pod *v1.Pod)
	OnAssumePod(pod *v1.Pod)
	OnForgetPod(pod *v1.Pod)
}

type gpuPredicateManager struct {
	cpu      PredicateManager // the existing implementation: EventsToRegister and every ask that is routed
	host     *C.ykhost_t
	lock     sync.Mutex // guards host against destroy; libykhost serialises its own entry points
	Counters GPUPredicateCounters
}

var _ PredicateManager = &gpuPredicateManager{}
var _ CacheObserver = &gpuPredicateManager{}

// NewGPUPredicateManager opens the device and returns the GPU-backed manager, or the CPU manager itself when the engine
// cannot be created (no GPU visible, library mismatch): the scheduler never starts without a working PredicateManager.
func NewGPUPredicateManager(cpu PredicateManager, device int) PredicateManager {
	errBuf := (*C.char)(C.malloc(512))
	defer C.free(unsafe.Pointer(errBuf))
	host := C.ykhost_create(C.int32_t(device), errBuf, 512)
	if host == nil {
		log.Log(log.ShimPredicates).Warn("GPU predicate engine unavailable, using the CPU predicate manager",
			zap.String("reason", C.GoString(errBuf)))
		return cpu
	}
	if C.ykpred_abi_version() != C.YKPRED_ABI_VERSION {
		log.Log(log.ShimPredicates).Warn("GPU predicate engine ABI mismatch, using the CPU predicate manager")
		C.ykhost_destroy(host)
		return cpu
	}
	log.Log(log.ShimPredicates).Info("GPU predicate engine created", zap.Int("device", device))
	return &gpuPredicateManager{cpu: cpu, host: host}
}

// NewConfiguredPredicateManager is what pkg/cache/context.go:130 calls instead of NewPredicateManager: the configuration
// key service.predicateEngine selects the implementation.
func NewConfiguredPredicateManager(handle fwk.Handle, engine string, device int) PredicateManager {
	cpu := NewPredicateManager(handle)
	if strings.EqualFold(engine, PredicateEngineGPU) {
		return NewGPUPredicateManager(cpu, device)
	}
	return cpu
}

// Close releases the device; the manager must not be used afterwards.
func (m *gpuPredicateManager) Close() {
	m.lock.Lock()
	defer m.lock.Unlock()
	if m.host != nil {
		C.ykhost_destroy(m.host)
		m.host = nil
	}
}

// EventsToRegister is pod-independent bookkeeping of the scheduling queue: unchanged (predicate_manager.go:70-132).
func (m *gpuPredicateManager) EventsToRegister(queueingHintFn fwk.QueueingHintFn) []fwk.ClusterEventWithHint {
	return m.cpu.EventsToRegister(queueingHintFn)
}

// Predicates keeps the reference contract (predicate_manager.go:134-139,206-219): ("", nil) when the pod fits, else the
// name of the first failing plugin ("" when a PreFilter plugin rejected the pod) and an error carrying the status message.
func (m *gpuPredicateManager) Predicates(pod *v1.Pod, node *framework.NodeInfo, allocate bool) (string, error) {
	if pod == nil || node == nil || node.Node() == nil {
		return m.cpu.Predicates(pod, node, allocate)
	}
	uid := C.CString(string(pod.UID))
	defer C.free(unsafe.Pointer(uid))
	name := C.CString(node.Node().Name)
	defer C.free(unsafe.Pointer(name))
	podIndex := C.ykhost_pod_index(m.host, uid)
	nodeIndex := C.ykhost_node_index(m.host, name)
	if podIndex < 0 || nodeIndex < 0 {
		m.Counters.RoutedNotMirrored.Add(1)
		return m.cpu.Predicates(pod, node, allocate)
	}
	var plugin [64]C.char
	var message [1024]C.char
	alloc := C.int32_t(0)
	if allocate {
		alloc = 1
	}
	rc := C.ykhost_predicates(m.host, podIndex, nodeIndex, alloc, &plugin[0], 64, &message[0], 1024)
	switch {
	case rc == 1:
		m.Counters.DeviceAnswers.Add(1)
		return "", nil
	case rc == 0:
		m.Counters.DeviceAnswers.Add(1)
		return C.GoString(&plugin[0]), errors.New(C.GoString(&message[0]))
	case rc == C.YKHOST_E_UNSUPPORTED:
		m.Counters.RoutedUnsupported.Add(1)
		return m.cpu.Predicates(pod, node, allocate)
	default:
		m.Counters.RoutedOnError.Add(1)
		log.Log(log.ShimPredicates).Warn("GPU predicate engine error, routing the call to the CPU predicate manager",
			zap.String("error", C.GoString(C.ykhost_last_error(m.host))))
		return m.cpu.Predicates(pod, node, allocate)
	}
}

// PreemptionPredicates keeps the reference contract (predicate_manager.go:141-179): the first index >= startIndex at
// which the pod fits once victims[0..index] are removed from the node, or -1.
func (m *gpuPredicateManager) PreemptionPredicates(pod *v1.Pod, node *framework.NodeInfo, victims []*v1.Pod, startIndex int) int {
	if pod == nil || node == nil || node.Node() == nil {
		return m.cpu.PreemptionPredicates(pod, node, victims, startIndex)
	}
	uid := C.CString(string(pod.UID))
	defer C.free(unsafe.Pointer(uid))
	name := C.CString(node.Node().Name)
	defer C.free(unsafe.Pointer(name))
	podIndex := C.ykhost_pod_index(m.host, uid)
	nodeIndex := C.ykhost_node_index(m.host, name)
	reason := make([]C.char, 600)
	if podIndex < 0 || nodeIndex < 0 || C.ykhost_ask_supported(m.host, podIndex, &reason[0], 600) != 1 {
		m.Counters.RoutedNotMirrored.Add(1)
		return m.cpu.PreemptionPredicates(pod, node, victims, startIndex)
	}
	// victim UIDs as a C array of C strings; a nil victim stays a NULL entry (predicate_manager.go:181-192 skips it)
	count := len(victims)
	slots := count
	if slots == 0 {
		slots = 1
	}
	array := (**C.char)(C.calloc(C.size_t(slots), C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(array))
	entries := unsafe.Slice(array, slots)
	for i, victim := range victims {
		if victim != nil {
			entries[i] = C.CString(string(victim.UID))
		}
	}
	defer func() {
		for _, entry := range entries {
			if entry != nil {
				C.free(unsafe.Pointer(entry))
			}
		}
	}()
	index := C.ykhost_preemption_predicates(m.host, podIndex, nodeIndex, array, C.int32_t(count), C.int32_t(startIndex))
	if index < -1 {
		m.Counters.RoutedOnError.Add(1)
		return m.cpu.PreemptionPredicates(pod, node, victims, startIndex)
	}
	m.Counters.PreemptionAnswers.Add(1)
	return int(index)
}

// ---- mirror maintenance (CacheObserver) ----------------------------------------------------------------------------------

func (m *gpuPredicateManager) marshal(object interface{}) *C.char {
	data, err := json.Marshal(object)
	if err != nil {
		m.Counters.MirrorErrors.Add(1)
		log.Log(log.ShimPredicates).Warn("unable to marshal object for the GPU predicate mirror", zap.Error(err))
		return nil
	}
	return C.CString(string(data))
}

func (m *gpuPredicateManager) mirrorResult(call string, rc C.int32_t) {
	m.Counters.MirrorUpdates.Add(1)
	if rc < 0 {
		m.Counters.MirrorErrors.Add(1)
		log.Log(log.ShimPredicates).Warn("GPU predicate mirror update failed",
			zap.String("call", call), zap.String("error", C.GoString(C.ykhost_last_error(m.host))))
	}
}

// OnUpdateNode: SchedulerCache.UpdateNode (scheduler_cache.go:148-187).
func (m *gpuPredicateManager) OnUpdateNode(node *v1.Node) {
	if text := m.marshal(node); text != nil {
		defer C.free(unsafe.Pointer(text))
		m.mirrorResult("ykhost_update_node", C.ykhost_update_node(m.host, text))
	}
}

// OnRemoveNode: SchedulerCache.RemoveNode (scheduler_cache.go:189-239).
func (m *gpuPredicateManager) OnRemoveNode(node *v1.Node) {
	name := C.CString(node.Name)
	defer C.free(unsafe.Pointer(name))
	m.mirrorResult("ykhost_remove_node", C.ykhost_remove_node(m.host, name))
}

// OnUpdatePod: SchedulerCache.UpdatePod (scheduler_cache.go:303-388).
func (m *gpuPredicateManager) OnUpdatePod(pod *v1.Pod) {
	if text := m.marshal(pod); text != nil {
		defer C.free(unsafe.Pointer(text))
		m.mirrorResult("ykhost_update_pod", C.ykhost_update_pod(m.host, text))
	}
}

// OnRemovePod: SchedulerCache.RemovePod (scheduler_cache.go:390-421).
func (m *gpuPredicateManager) OnRemovePod(pod *v1.Pod) {
	uid := C.CString(string(pod.UID))
	defer C.free(unsafe.Pointer(uid))
	m.mirrorResult("ykhost_remove_pod", C.ykhost_remove_pod(m.host, uid))
}

// OnAssumePod: SchedulerCache.AssumePod (scheduler_cache.go:443-461); the pod carries the node it was assumed on.
func (m *gpuPredicateManager) OnAssumePod(pod *v1.Pod) {
	uid := C.CString(string(pod.UID))
	defer C.free(unsafe.Pointer(uid))
	name := C.CString(pod.Spec.NodeName)
	defer C.free(unsafe.Pointer(name))
	m.mirrorResult("ykhost_assume_pod", C.ykhost_assume_pod(m.host, uid, name))
}

// OnForgetPod: SchedulerCache.ForgetPod (scheduler_cache.go:463-484).
func (m *gpuPredicateManager) OnForgetPod(pod *v1.Pod) {
	uid := C.CString(string(pod.UID))
	defer C.free(unsafe.Pointer(uid))
	m.mirrorResult("ykhost_forget_pod", C.ykhost_forget_pod(m.host, uid))
}

// ReplayState fills the mirror in bulk: every node, then every pod, each kind in ONE cgo crossing
// (ykhost_update_nodes_batch / ykhost_update_pods_batch). It is what Context.InitializeState (pkg/cache/context.go:1411-1484)
// amounts to for the mirror when the informer caches are replayed at start-up: the per-object hooks above do the same work
// with one crossing, one lock round trip and one C string per object. Pods of one Deployment / task group differ in name
// and uid only; libykhost recognises their template text and parses it once.
func (m *gpuPredicateManager) ReplayState(nodes []*v1.Node, pods []*v1.Pod) error {
	join := func(count int, object func(int) interface{}) (*C.char, C.int64_t, error) {
		var buffer strings.Builder
		for i := 0; i < count; i++ {
			data, err := json.Marshal(object(i))
			if err != nil {
				return nil, 0, err
			}
			buffer.Write(data)
			buffer.WriteByte('\n')
		}
		return C.CString(buffer.String()), C.int64_t(buffer.Len()), nil
	}
	text, length, err := join(len(nodes), func(i int) interface{} { return nodes[i] })
	if err != nil {
		return err
	}
	applied := C.ykhost_update_nodes_batch(m.host, text, length)
	C.free(unsafe.Pointer(text))
	if int(applied) != len(nodes) {
		m.Counters.MirrorErrors.Add(1)
		return fmt.Errorf("ykhost_update_nodes_batch: %s", C.GoString(C.ykhost_last_error(m.host)))
	}
	text, length, err = join(len(pods), func(i int) interface{} { return pods[i] })
	if err != nil {
		return err
	}
	applied = C.ykhost_update_pods_batch(m.host, text, length)
	C.free(unsafe.Pointer(text))
	if int(applied) != len(pods) {
		m.Counters.MirrorErrors.Add(1)
		return fmt.Errorf("ykhost_update_pods_batch: %s", C.GoString(C.ykhost_last_error(m.host)))
	}
	m.Counters.MirrorUpdates.Add(int64(len(nodes) + len(pods)))
	return nil
}

// ---- batched evaluation ------------------------------------------------------------------------------------------------------

// Refresh brings the device-resident P x N feasibility bitmap, the per-ask feasible counts and the per-ask bin-pack
// decisions up to date with the mirror: only the bitmap columns of changed nodes and the rows of changed asks are
// re-evaluated when that is possible (ykhost_evaluate_dirty), a full pass otherwise. Returns the number of node columns that
// were patched, or -1 after a full pass. A scheduling cycle that wants candidates for every pending ask calls this once and
// reads the results through Layout(); the per-pair Predicates() callback does not depend on it.
func (m *gpuPredicateManager) Refresh(allocate bool) (int, error) {
	alloc := C.int32_t(0)
	if allocate {
		alloc = 1
	}
	var patched C.int32_t
	options := C.uint32_t(C.YKPRED_OUT_BITMAP | C.YKPRED_OUT_COUNTS | C.YKPRED_OUT_DECISIONS)
	if rc := C.ykhost_evaluate_dirty(m.host, alloc, options, &patched); rc < 0 {
		return 0, fmt.Errorf("ykhost_evaluate_dirty: %s", C.GoString(C.ykhost_last_error(m.host)))
	}
	return int(patched), nil
}

// BitmapLayout describes the device-resident results of the last Refresh (device pointers: for consumers on the GPU side,
// e.g. a batched scheduler-interface callback; host code reads single answers through Predicates()).
type BitmapLayout struct {
	Nodes, Asks, RowWords, RowStride, Rows int
	// Bitmap rows are permuted for the writer: bit (n & 63) of word RowOfPod[p]*RowStride + (n >> 6) says whether ask p fits node n.
	Bitmap, RowOfPod, Counts, Decisions unsafe.Pointer
}

// Layout returns where the results of the last Refresh live.
func (m *gpuPredicateManager) Layout() (BitmapLayout, error) {
	var layout C.ykpred_layout_t
	if rc := C.ykpred_get_layout(C.ykhost_engine(m.host), &layout); rc != C.YKPRED_OK {
		return BitmapLayout{}, errors.New("ykpred_get_layout failed")
	}
	return BitmapLayout{
		Nodes: int(layout.num_nodes), Asks: int(layout.num_pods), RowWords: int(layout.row_words), RowStride: int(layout.row_stride),
		Rows: int(layout.num_rows), Bitmap: layout.bitmap, RowOfPod: layout.row_of_pod, Counts: layout.counts, Decisions: layout.decisions,
	}, nil
}

// Candidates returns the first k feasible nodes of an ask in bin-pack order (ascending score, ties by NodeID) — the node
// loop the core runs through Predicates() callbacks, answered from the resident bitmap row in one call. It needs a Refresh
// with nothing changed since; (nil, false) tells the caller to fall back to the per-node callbacks.
func (m *gpuPredicateManager) Candidates(pod *v1.Pod, k int, allocate bool) ([]int32, bool) {
	if pod == nil || k <= 0 {
		return nil, false
	}
	uid := C.CString(string(pod.UID))
	defer C.free(unsafe.Pointer(uid))
	podIndex := C.ykhost_pod_index(m.host, uid)
	if podIndex < 0 {
		return nil, false
	}
	alloc := C.int32_t(0)
	if allocate {
		alloc = 1
	}
	nodes := make([]C.int32_t, k)
	found := C.ykhost_candidates(m.host, podIndex, alloc, C.int32_t(k), &nodes[0])
	if found < 0 {
		return nil, false
	}
	out := make([]int32, int(found))
	for i := range out {
		out[i] = int32(nodes[i])
	}
	return out, true
}

// ResidentStats: Predicates() calls answered from the mirrored resident answer, per pair because the node's column changed,
// by whole-ask device queries, and the answer / failing-plugin fetches behind them.
func (m *gpuPredicateManager) ResidentStats() (resident, dirtyColumn, query, answerFetches, codeFetches int64) {
	var out [5]C.int64_t
	C.ykhost_resident_stats(m.host, &out[0])
	return int64(out[0]), int64(out[1]), int64(out[2]), int64(out[3]), int64(out[4])
}

// RoutingStats: asks marked unsupported at the last encode, Predicates() calls the engine refused, dictionary growths.
func (m *gpuPredicateManager) RoutingStats() (unsupported, refused, growths int64) {
	var out [3]C.int64_t
	C.ykhost_routing_stats(m.host, &out[0])
	return int64(out[0]), int64(out[1]), int64(out[2])
}

// AllocateRound decides a scheduling round with conflict-resolved decisions: pods[i] is decided with pods[0..i-1] assumed on
// their nodes — what the core's allocation loop obtains by calling Predicates() down the bin-pack order and AssumePod after every
// allocation (scheduler_callback.go:49-98, context.go:828-885), in ONE device call: node resources, pod slots, host ports and
// the PreFilter state of PodTopologySpread / InterPodAffinity are kept live on the device while the round runs
// (ykhost_allocate_round). nodes[i] is the node index in the mirror's node order (NodeName resolves it), -1 when no node fits, -2
// when the ask is routed to the CPU manager. On a node-sharded cluster the call is collective (every rank's manager calls it with
// the same pods) and nodes[i] is the index in the whole cluster: shard offset + local index (ykpred_comm_info). apply = true
// marks the chosen nodes' asks assumed in the mirror exactly as OnAssumePod would; with apply = false the core confirms through
// OnAssumePod itself. Needs a current Refresh(allocate) with decisions.
func (m *gpuPredicateManager) AllocateRound(pods []*v1.Pod, apply bool) ([]int32, error) {
	if len(pods) == 0 {
		return nil, nil
	}
	asks := make([]C.int32_t, len(pods))
	for i, pod := range pods {
		uid := C.CString(string(pod.UID))
		asks[i] = C.ykhost_pod_index(m.host, uid)
		C.free(unsafe.Pointer(uid))
		if asks[i] < 0 {
			return nil, fmt.Errorf("pod %s is not a pending ask of the mirror", pod.UID)
		}
	}
	doApply := C.int32_t(0)
	if apply {
		doApply = 1
	}
	nodes := make([]C.int32_t, len(pods))
	if rc := C.ykhost_allocate_round(m.host, C.int32_t(len(pods)), &asks[0], doApply, &nodes[0]); rc < 0 {
		return nil, fmt.Errorf("ykhost_allocate_round: %s", C.GoString(C.ykhost_last_error(m.host)))
	}
	out := make([]int32, len(pods))
	for i := range out {
		out[i] = int32(nodes[i])
	}
	return out, nil
}

// RoundStats: rounds decided by one device call, asks decided in them, asks decided ask by ask, asks routed to the CPU manager.
func (m *gpuPredicateManager) RoundStats() (deviceRounds, deviceAsks, oneByOne, routed int64) {
	var out [4]C.int64_t
	C.ykhost_round_stats(m.host, &out[0])
	return int64(out[0]), int64(out[1]), int64(out[2]), int64(out[3])
}

// DeviceErrors: engine calls that failed (lost device, failed allocation) since the manager was created. Each one marked the
// whole device state stale; the next Refresh re-uploads. A growing counter is what an operator alerts on.
func (m *gpuPredicateManager) DeviceErrors() int64 {
	return int64(C.ykhost_device_errors(m.host))
}

// IngestTiming: where the time of the pod batches went (ReplayState): scanning threads of the last batch, microseconds of the
// parallel scan, microseconds of the cache pass, batches cut into pieces, batches whose cache pass was the bulk pass on every core.
func (m *gpuPredicateManager) IngestTiming() (threads, scanMicros, cacheMicros, parallelBatches, bulkBatches int64) {
	var out [5]C.int64_t
	C.ykhost_ingest_timing(m.host, &out[0])
	return int64(out[0]), int64(out[1]), int64(out[2]), int64(out[3]), int64(out[4])
}
