/*
 * mi_convnet.h -- C ABI of the MI355X (gfx950) conv-net compute core.
 *
 * This is the drop-in boundary between a cuda-convnet style layer graph and the
 * HIP kernels: one entry point per public function of the reference's kernel
 * headers (include/cudaconv2/cudaconv2.cuh, include/cudaconv2/conv_util.cuh,
 * include/layer_kernels.cuh), the NVMatrix verbs the layers use
 * (include/nvmatrix/nvmatrix.cuh) and the SGD update of include/weights.cuh.
 * Every entry cites the reference interface it replaces.
 *
 * Conventions
 *  - plain C: raw device pointers, explicit dimensions, a stream handle
 *    (hipStream_t passed as void*; NULL = the default stream).  No torch types.
 *  - every function returns 0 on success, a negative MCN_E* code otherwise and
 *    never aborts; mcn_last_error() returns a message for the calling thread.
 *  - no hidden allocation: routines that need scratch take (workspace,
 *    workspaceBytes) and have a *_workspace_bytes() companion.
 *  - layouts are the reference's: activations are row-major
 *    (channels*rows*cols, numImages), images contiguous; conv filters
 *    (filterChannels*filterPixels, numFilters); local (unshared) filters
 *    (modules*filterChannels*filterPixels, numFilters)  (filter_acts.cu:39-43).
 *  - all data is fp32; labels are fp32-encoded integers (layer_kernels.cu:64).
 *  - unlike the reference (square only: layer.py:351-354) every spatial routine
 *    takes Y and X sizes separately.
 *  - "targets = scaleTargets*targets + scaleOutput*result"; scaleTargets == 0
 *    means targets is write-only (never read, may hold NaNs).
 */
#ifndef MI_CONVNET_H
#define MI_CONVNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCN_OK 0
#define MCN_EINVAL (-1)   /* bad argument / unsupported shape            */
#define MCN_EHIP (-2)     /* a HIP runtime call failed                   */
#define MCN_EWORKSPACE (-3) /* workspace too small                       */
#define MCN_ERCCL (-4)    /* RCCL missing or a collective failed         */

/* neuron types, neuron.cu:32-75 (makeNeuron) */
enum mcn_neuron {
    MCN_NEURON_IDENT = 0, MCN_NEURON_RELU = 1, MCN_NEURON_LOGISTIC = 2, MCN_NEURON_BRELU = 3,
    MCN_NEURON_ABS = 4, MCN_NEURON_TANH = 5, MCN_NEURON_SOFTRELU = 6, MCN_NEURON_SQUARE = 7,
    MCN_NEURON_SQRT = 8, MCN_NEURON_LINEAR = 9
};

/* ---- runtime (replaces cudaSetDevice/cublasAlloc/cublasSetMatrix... nvmatrix.cu:60-170, convnet.cu:139-145) */
const char* mcn_last_error(void);
/* kernel launches this thread has made through the library so far (profilers use it to drop spans that launched nothing) */
size_t mcn_launch_count(void);
/* ... and the host nanoseconds this thread has spent inside the runtime's kernel-launch call (submission cost, as opposed
   to time a submitting thread spends blocked; no counterpart in the reference, which never measures its host side) */
uint64_t mcn_launch_ns(void);
int mcn_device_count(int* count);
int mcn_set_device(int device);
int mcn_alloc(void** ptr, size_t bytes);
int mcn_free(void* ptr);
int mcn_host_alloc(void** ptr, size_t bytes);          /* pinned host memory */
int mcn_host_free(void* ptr);
int mcn_memset(void* ptr, int value, size_t bytes, void* stream);
int mcn_h2d(void* dst, const void* src, size_t bytes, void* stream);
int mcn_d2h(void* dst, const void* src, size_t bytes, void* stream);
int mcn_d2d(void* dst, const void* src, size_t bytes, void* stream);
/* kind: 0 h2d, 1 d2h, 2 d2d.  Column-slice copies (data.cu:72-98 sliceCols). */
int mcn_copy2d(void* dst, size_t dpitchBytes, const void* src, size_t spitchBytes, size_t widthBytes,
               size_t height, int kind, void* stream);
/* All minibatch column slices of a resident (rows, numCases) batch at once (data.cu:72-98 run ahead of time):
 * dst holds, minibatch after minibatch, a dense (rows, n_i) matrix with n_i = min(minibatch, numCases - i*minibatch);
 * slice i starts at dst + i*minibatch*rows. */
int mcn_tile_minibatches(const float* src, float* dst, int rows, size_t numCases, int minibatch, void* stream);
int mcn_stream_create(void** stream);
int mcn_stream_destroy(void* stream);
/* A stream of another priority level (level < 0 lowest, > 0 highest): the runtime hands out its hardware queues per level and
 * streams of one level share them once there are more streams than queues, so a background stream of its own level (the
 * upload stream) leaves the normal level's queues to the streams that must run side by side. */
int mcn_stream_create_priority(void** stream, int level);
int mcn_stream_sync(void* stream);
int mcn_device_sync(void);
int mcn_event_create(void** event);
int mcn_event_create_sync(void** event);               /* no timing: for stream ordering only */
int mcn_event_destroy(void* event);
int mcn_event_record(void* event, void* stream);
int mcn_event_sync(void* event);
int mcn_stream_wait_event(void* stream, void* event);
/* hipEventQuery: 1 = complete, 0 = not yet, < 0 = error */
int mcn_event_query(void* event);
int mcn_event_elapsed_ms(float* ms, void* start, void* stop);

/* Stream ticks (no counterpart in the reference): mcn_stream_tick queues a one-thread kernel that stores `value` into
 * `flag`, 64 bytes of host-visible coherent memory from mcn_tick_alloc.  A host thread that polls *flag learns that the
 * stream has passed this point ~5 us before mcn_event_query reports an event recorded at the same place, and the stream
 * loses 1.5 us instead of 3 us (tools/event_poll_probe.hip).  Values should increase; compare with (int)(*flag - value) >= 0. */
int mcn_tick_alloc(int** flag);
int mcn_tick_free(int* flag);
int mcn_stream_tick(int* flag, int value, void* stream);
/* The same tick carried by the calling thread's NEXT launch on `stream` instead of a kernel of its own: kernels with a tick
 * slot (mcn_gemm's classifier forms, mcn_fc_bprop_acts_multi, the imgActs family, the strip form of
 * mcn_conv_weight_acts_reduced) store the value in their first instructions -- a kernel starts only when everything queued
 * before it on its stream has finished -- and any other launch is followed by a one-thread tick kernel.  mcn_flush_tick sends
 * an armed tick at once; call it before blocking on whoever polls for it. */
int mcn_arm_tick(int* flag, int value, void* stream);
/* ... and the same value into gate[0] (device memory, mcn_gate_alloc) as well: what a stream gate on another stream waits
 * for (mcn_stream_gate) -- a fork that needs neither a host thread's poll nor a barrier packet */
int mcn_arm_tick_gate(int* flag, int* gate, int value, void* stream);
int mcn_flush_tick(void);
/* Stream gates (no counterpart in the reference): a dependency between two streams of ONE device that never passes through
 * the command processor's barrier packets.  mcn_stream_signal queues a one-thread kernel on the producing stream that
 * stores `value` (release, device scope) into gate[0]; mcn_stream_gate queues a one-thread kernel on the consuming stream
 * that spins until (int)(gate[0] - value) >= 0 and then ends, so that everything queued behind it starts.  Measured on
 * MI355X / ROCm 7.2: a hipStreamWaitEvent whose event has not fired yet holds the waiting queue ~10-15 us after it fires;
 * a gate holds it for the store to become visible plus one kernel dispatch (tools/step_timeline.py).
 * A gate that sees nothing for `timeoutMs` gives up, stores 1 into gate[1] (mcn_gate_timed_out) and ends: the grid always
 * drains.  Gates need the two streams to make progress independently (separate hardware queues); mcn_gate_selftest
 * queues a gate on `waiter` and then its signal on `signaller` and reports whether the gate was released by the signal
 * (1) or ran into its timeout (0: use events).  gate: 64 bytes of device memory from mcn_gate_alloc, zero-initialised. */
int mcn_gate_alloc(int** gate);
int mcn_gate_free(int* gate);
int mcn_stream_signal(int* gate, int value, void* stream);
int mcn_stream_gate(int* gate, int value, int timeoutMs, void* stream);
int mcn_gate_timed_out(int* gate, int* timedOut, void* stream); /* blocking read of gate[1] */
int mcn_gate_selftest(int* gate, void* waiter, void* signaller, int* concurrent);

/* Step graphs (no counterpart in the reference, whose step is ~45 launches and several blocking copies per minibatch):
 * everything issued on `stream`, and on streams that join it through events, between begin and end is captured into one
 * executable graph (nothing runs during capture); mcn_graph_launch replays it on a stream.  numNodes may be NULL. */
int mcn_graph_begin_capture(void* stream);
int mcn_graph_end_capture(void* stream, void** graphExec, int* numNodes);
int mcn_graph_launch(void* graphExec, void* stream);
int mcn_graph_destroy(void* graphExec);

/* ---- cudaconv2.cuh:36-78 ------------------------------------------------------------------ */
/* convFilterActs (cudaconv2.cuh:39-42, filter_acts.cu:1188-1193) / localFilterActs (:47-50, :1201-1206).
 * images (numImgColors*imgSizeY*imgSizeX, numImages) with row stride imgStride (>= numImages);
 * targets (numFilters*numModulesY*numModulesX, numImages).  paddingStart <= 0. */
int mcn_conv_filter_acts(const float* images, const float* filters, float* targets,
                         int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                         int numFilters, int filterSize, int numModulesY, int numModulesX,
                         int paddingStart, int moduleStride, int numGroups,
                         float scaleTargets, float scaleOutput, void* stream);
int mcn_local_filter_acts(const float* images, const float* filters, float* targets,
                          int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                          int numFilters, int filterSize, int numModulesY, int numModulesX,
                          int paddingStart, int moduleStride, int numGroups,
                          float scaleTargets, float scaleOutput, void* stream);
/* Fused forward used by the layer graph: filterActs, then (+ biases) and the
 * neuron in the epilogue (ConvLayer::fpropActs layer.cu:443-462 + NeuronLayer
 * layer.cu:249 when the neuron shares the conv's activity matrix).  biases may
 * be NULL; biasPerModule = 0: one bias per filter (sharedBiases), 1: one per
 * (filter, module).  Requires scaleTargets == 0 semantics (targets write-only). */
int mcn_filter_acts_bias_neuron(const float* images, const float* filters, const float* biases, float* targets,
                                int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                int numFilters, int filterSize, int numModulesY, int numModulesX,
                                int paddingStart, int moduleStride, int numGroups, int conv,
                                int biasPerModule, int neuron, float neuronA, float neuronB, void* stream);
/* The same launch also writing the max pool that follows the layer (PoolLayer::fpropActs layer.cu:718-726 = convLocalPool with
 * MaxPooler, conv_util.cuh, for sizeX 2, stride 2, start 0): targets as above, poolTargets (numFilters*(numModulesY/2)*(numModulesX/2),
 * numImages).  Shared-weight conv layers whose modules tile the image exactly in an even number of rows and columns
 * (mcn_filter_acts_pool_supported = 1); both outputs are bitwise those of mcn_filter_acts_bias_neuron + mcn_local_pool. */
int mcn_filter_acts_pool_supported(const float* images, const float* filters, const float* biases, const float* targets,
                                   int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                   int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                   int numGroups, int conv, int biasPerModule, int neuron);
int mcn_filter_acts_bias_neuron_pool(const float* images, const float* filters, const float* biases, float* targets,
                                     float* poolTargets, int numImages, int imgStride, int imgSizeY, int imgSizeX,
                                     int numImgColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                     int paddingStart, int moduleStride, int numGroups, int conv, int biasPerModule, int neuron,
                                     float neuronA, float neuronB, void* stream);

/* convImgActs (cudaconv2.cuh:54-56, img_acts.cu:1892-1900) / localImgActs (:60-62, :1904-1912).
 * hidActs (numFilters*modules, numImages) -> targets (numImgColors*imgSizeY*imgSizeX, numImages). */
int mcn_conv_img_acts(const float* hidActs, const float* filters, float* targets,
                      int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                      int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                      int numGroups, float scaleTargets, float scaleOutput, void* stream);
int mcn_local_img_acts(const float* hidActs, const float* filters, float* targets,
                       int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                       int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                       int numGroups, float scaleTargets, float scaleOutput, void* stream);

/* imgActs + MaxPoolLayer::bpropActs (layer.cu:734, convLocalMaxUndo conv_util.cuh) + NeuronLayer::bpropActs (layer.cu:251)
 * in one launch, for a conv / local layer (conv = 1 / 0) whose input is a max pool with sizeX 2, stride 2, start 0 over maps of
 * exactly (2 imgSizeY, 2 imgSizeX) pixels (mcn_img_acts_max_undo_supported): the gradient of each pool output goes from the
 * accumulators straight to the pool input it came from,
 *   undoTarget[c, 2y+i, 2x+j, n] = neuronGrad( (poolInput[c,2y+i,2x+j,n] == poolOutput[c,y,x,n]) * g[c,y,x,n] ),
 *   g = scaleTargets*poolGrads + scaleOutput*imgActs(hidActs, filters)
 * -- bit for bit what the three separate calls leave in undoTarget.  poolGrads (C*imgSizeY*imgSizeX, numImages) is only read,
 * and only when scaleTargets != 0 (may be NULL otherwise); neuron: -1 (none), MCN_NEURON_IDENT or MCN_NEURON_RELU. */
int mcn_img_acts_max_undo_supported(int numImages, int poolSizeX, int poolStart, int poolStride, int poolOutY, int poolOutX,
                                    int poolImgY, int poolImgX, int neuron);
int mcn_img_acts_max_undo(const float* hidActs, const float* filters, const float* poolGrads, int numImages, int imgSizeY,
                          int imgSizeX, int numImgColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                          int paddingStart, int moduleStride, int numGroups, int conv, float scaleTargets, float scaleOutput,
                          const float* poolInput, const float* poolOutput, float* undoTarget, int neuron, float neuronA,
                          float neuronB, void* stream);

/* Tuning switches of the calling thread (no counterpart in the reference).  "img_acts_wide": 4 (default) | 2 | 0 -- waves per
 * input pixel in the backward-data product of wide layers (>= 64 filters per group, an even number of 16-channel tiles,
 * few pixels x images).  4 is the fastest launch on an idle GPU (DeepID2+ conv2 120 -> 85 us); a caller that runs weight
 * gradients on a second stream at the same time sets 0 (see conv_bwd_data.hip). */
int mcn_set_thread_option(const char* name, int value);

/* convWeightActs (cudaconv2.cuh:67-70, weight_acts.cu:955-964): targets
 * ((modules/partialSum)*filterChannels*filterPixels, numFilters); partialSum <= 0 means all modules.
 * localWeightActs (cudaconv2.cuh:76-78, weight_acts.cu:966-976): targets
 * (modules*filterChannels*filterPixels, numFilters). */
int mcn_conv_weight_acts(const float* images, const float* hidActs, float* targets,
                         int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                         int numFilters, int filterSize, int numModulesY, int numModulesX,
                         int paddingStart, int moduleStride, int numGroups, int partialSum,
                         float scaleTargets, float scaleOutput, void* stream);
int mcn_local_weight_acts(const float* images, const float* hidActs, float* targets,
                          int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                          int numFilters, int filterSize, int numModulesY, int numModulesX,
                          int paddingStart, int moduleStride, int numGroups,
                          float scaleTargets, float scaleOutput, void* stream);
/* ConvLayer::bpropWeights (layer.cu:476-496) in one call: weight gradient summed
 * over ALL modules, grad = scaleTargets*grad + scaleOutput*sum.  Replaces the
 * reference's (modules/partialSum) x weights temporary + addSum with a private
 * module split and a fixed-order second pass over `workspace`. */
size_t mcn_conv_weight_acts_reduced_workspace_bytes(int numImgColors, int numFilters, int filterSize,
                                                    int numModulesY, int numModulesX, int numGroups);
int mcn_conv_weight_acts_reduced(const float* images, const float* hidActs, float* grad,
                                 void* workspace, size_t workspaceBytes,
                                 int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                 int numFilters, int filterSize, int numModulesY, int numModulesX,
                                 int paddingStart, int moduleStride, int numGroups,
                                 float scaleTargets, float scaleOutput, void* stream);

/* First stage only of mcn_conv_weight_acts_reduced (the reference's own first stage: convWeightActs into its partialSum
 * temporary, layer.cu:476-485): `partials` receives *numChunks slabs of (filterChannels*fs*fs x numFilters) floats, each
 * scaled by scaleOutput, whose sum over the slabs is the weight gradient; partialsBytes >=
 * mcn_conv_weight_acts_reduced_workspace_bytes(...).  For a consumer that sums them itself (mcn_sgd_plan_build_fused).
 * *numChunks depends only on the layer geometry and on whether numImages is a multiple of 32 (and 16-byte alignment):
 * mcn_conv_weight_acts_partials_count returns it without launching anything.
 * biasPartials (may be NULL): where the kernel form allows it (*biasDone = 1; *biasCapable of the count query) the launch
 * also leaves the shared-bias gradient (ConvLayer::bpropBiases, layer.cu:464-474) in pieces, *numChunks slabs of numFilters
 * floats, each biasScale times the sum of the gradient rows over the slab's modules and all images -- it reads those rows
 * anyway, and the separate row sum over the whole gradient matrix need not run. */
int mcn_conv_weight_acts_partials_count(int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                        int filterSize, int numModulesY, int numModulesX, int moduleStride, int numGroups,
                                        int* biasCapable);
int mcn_conv_weight_acts_partials(const float* images, const float* hidActs, void* partials, size_t partialsBytes,
                                  int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                                  int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                  int numGroups, float scaleOutput, int* numChunks, float* biasPartials, float biasScale,
                                  int* biasDone, void* stream);
/* No counterpart in the reference.  mcn_conv_weight_acts_reduced for callers that cannot allocate device memory or
 * bring a workspace (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the
 * module-chunk partials live in a static device array of mcn_conv_weight_acts_reduced_static_max_bytes() = 4 MiB
 * that each GPU the library runs on holds until the process ends.  Shapes whose partials need more return
 * MCN_EWORKSPACE (every DeepID / DeepID2+ / CIFAR conv layer at minibatch 128 fits).  Calls no hipMalloc / hipFree.
 * All calls on one GPU share the array: calls that overlap on one GPU (two streams) race. */
size_t mcn_conv_weight_acts_reduced_static_max_bytes(void);
int mcn_conv_weight_acts_reduced_static(const float* images, const float* hidActs, float* grad,
                                        int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                        int numFilters, int filterSize, int numModulesY, int numModulesX,
                                        int paddingStart, int moduleStride, int numGroups,
                                        float scaleTargets, float scaleOutput, void* stream);

/* ---- random-sparse connectivity (cudaconv2.cuh:80-124) ----------------------------------------------
 * Filter group g reads the numFilterColors image channels colorIndices[g*numFilterColors ..] (device ints; a concatenation
 * of overSample = numGroups*numFilterColors/numImgColors permutations of the image channels, layer.py:830-853).  filters
 * are (numFilterColors*filterPixels, numFilters) [conv] / (modules*numFilterColors*filterPixels, numFilters) [local], as in
 * the reference.  The channel table is applied once per call by a streaming pass into `workspace`, the product is the dense
 * grouped kernel over the re-ordered tensor (sparse_conv.hip).
 * conv/localFilterActsSparse (cudaconv2.cuh:80-94, filter_acts.cu:1221-1444) and conv/localWeightActsSparse
 * (cudaconv2.cuh:96-108): workspace >= mcn_sparse_gathered_bytes().
 * conv/localImgActsSparse (cudaconv2.cuh:110-124, img_acts.cu:1929-2583): workspace >= mcn_sparse_img_acts_workspace_bytes();
 * targets is (numImgColors*imgPixels, numImages) = scaleTargets*targets + scaleOutput * (sum over the overSample copies) --
 * the reference writes the (overSample, numImgColors, imgPixels, numImages) copies and its layer sums them (layer.cu:500-508,
 * where the sum overwrites the target even when scaleTargets == 1). */
size_t mcn_sparse_gathered_bytes(int numImages, int imgSizeY, int imgSizeX, int numFilterColors, int numGroups);
size_t mcn_sparse_img_acts_workspace_bytes(int numImages, int imgSizeY, int imgSizeX, int numFilterColors, int numGroups);
int mcn_conv_filter_acts_sparse(const float* images, const float* filters, float* targets, const int* colorIndices, void* workspace,
                                size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                int paddingStart, int moduleStride, int numGroups, float scaleTargets, float scaleOutput, void* stream);
int mcn_local_filter_acts_sparse(const float* images, const float* filters, float* targets, const int* colorIndices, void* workspace,
                                 size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                 int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                 int paddingStart, int moduleStride, int numGroups, float scaleTargets, float scaleOutput, void* stream);
int mcn_conv_img_acts_sparse(const float* hidActs, const float* filters, float* targets, const int* colorIndices, void* workspace,
                             size_t workspaceBytes, int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilterColors,
                             int numFilters, int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                             int numGroups, float scaleTargets, float scaleOutput, void* stream);
int mcn_local_img_acts_sparse(const float* hidActs, const float* filters, float* targets, const int* colorIndices, void* workspace,
                              size_t workspaceBytes, int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilterColors,
                              int numFilters, int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                              int numGroups, float scaleTargets, float scaleOutput, void* stream);
int mcn_conv_weight_acts_sparse(const float* images, const float* hidActs, float* targets, const int* colorIndices, void* workspace,
                                size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                int paddingStart, int moduleStride, int numGroups, int partialSum, float scaleTargets,
                                float scaleOutput, void* stream);
int mcn_local_weight_acts_sparse(const float* images, const float* hidActs, float* targets, const int* colorIndices, void* workspace,
                                 size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                 int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                 int paddingStart, int moduleStride, int numGroups, float scaleTargets, float scaleOutput,
                                 void* stream);
/* The passes themselves, for a layer graph that keeps the re-ordered tensor between forward and the weight gradient
 * (no counterpart in the reference, whose kernels chase colorIndices in their inner loops, filter_acts.cu:401-560):
 * target[k] = images[colorIndices[k]] for k < numIndices (a channel = imgPixels rows of numImages);
 * inverse[s*numImgColors + colorIndices[s*numImgColors + j]] = s*numImgColors + j;
 * target[c] = scaleTargets*target[c] + scaleOutput * sum_s gathered[inverse[s*numImgColors + c]]. */
int mcn_gather_channels(const float* images, int imgStride, const int* colorIndices, float* target, int numImages, int numImgColors,
                        int numIndices, int imgPixels, void* stream);
int mcn_invert_channel_indices(const int* colorIndices, int* inverse, int numImgColors, int overSample, void* stream);
int mcn_scatter_channels_sum(const float* gathered, const int* inverse, float* target, int numImages, int numImgColors, int overSample,
                             int imgPixels, float scaleTargets, float scaleOutput, void* stream);

/* ---- conv_util.cuh ---------------------------------------------------------------------- */
/* convLocalPool<MaxPooler|AvgPooler> (conv_util.cuh:327-517). mode 0 = max, 1 = avg (clipped window). */
int mcn_local_pool(const float* images, float* target, int numImages, int numFilters, int imgSizeY,
                   int imgSizeX, int subsX, int startX, int strideX, int outputsY, int outputsX, int mode,
                   void* stream);
/* convLocalMaxUndo (conv_util.cuh:40-41, conv_util.cu:1869-1958) */
int mcn_local_max_undo(const float* images, const float* maxGrads, const float* maxActs, float* target,
                       int numImages, int numFilters, int imgSizeY, int imgSizeX, int subsX, int startX,
                       int strideX, int outputsY, int outputsX, float scaleTargets, float scaleOutput,
                       void* stream);
/* convLocalMaxUndo followed by the gradient of the neuron that produced `images`
 * in place (NeuronLayer::bpropActs layer.cu:245 with actsGradTarget sharing):
 * target = neuronGrad(maxUndo(...), images).  Only neurons whose gradient needs
 * their output alone (ident, relu, brelu, logistic, tanh, sqrt, linear). */
int mcn_local_max_undo_neuron(const float* images, const float* maxGrads, const float* maxActs, float* target,
                              int numImages, int numFilters, int imgSizeY, int imgSizeX, int subsX, int startX,
                              int strideX, int outputsY, int outputsX, int neuron, float neuronA, float neuronB,
                              void* stream);
/* convLocalAvgUndo (conv_util.cuh:37-39, conv_util.cu:1968-2062) */
int mcn_local_avg_undo(const float* avgGrads, float* target, int numImages, int numFilters, int imgSizeY,
                       int imgSizeX, int subsX, int startX, int strideX, int outputsY, int outputsX,
                       float scaleTargets, float scaleOutput, void* stream);
/* convResponseNorm (conv_util.cuh:43, conv_util.cu:2064-2072) */
int mcn_response_norm(const float* images, float* denoms, float* target, int numImages, int numFilters,
                      int imgSizeY, int imgSizeX, int sizeX, float addScale, float powScale, void* stream);
/* convContrastNorm (conv_util.cuh:46-47, conv_util.cu:2074-2160): denoms = 1 + addScale * window sum of meanDiffs^2,
 * target = images * denoms^-powScale.  Its gradient convContrastNormUndo (conv_util.cu:2214-2217) is
 * mcn_response_norm_undo with inputs = meanDiffs. */
int mcn_contrast_norm(const float* images, const float* meanDiffs, float* denoms, float* target, int numImages,
                      int numFilters, int imgSizeY, int imgSizeX, int sizeX, float addScale, float powScale, void* stream);
/* convResponseNormUndo (conv_util.cuh:44-45, conv_util.cu:2228-2412). OVERWRITES acts. */
int mcn_response_norm_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts,
                           float* target, int numImages, int numFilters, int imgSizeY, int imgSizeX, int sizeX,
                           float addScale, float powScale, float scaleTargets, float scaleOutput, void* stream);
/* convResponseNormCrossMap (conv_util.cuh:69-70, conv_util.cu:2892-2894) */
int mcn_response_norm_cross_map(const float* images, float* denoms, float* target, int numImages,
                                int numFilters, int imgPixels, int sizeF, float addScale, float powScale,
                                int blocked, void* stream);
/* convResponseNormCrossMapUndo (conv_util.cuh:67-68, conv_util.cu:2801-2890). OVERWRITES acts. */
int mcn_response_norm_cross_map_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts,
                                     float* target, int numImages, int numFilters, int imgPixels, int sizeF,
                                     float addScale, float powScale, int blocked, float scaleTargets,
                                     float scaleOutput, void* stream);

/* ---- neuron.cuh ----------------------------------------------------------------------------- */
/* Neuron::activate (neuron.cuh:87-92); in-place allowed (out == in). */
int mcn_neuron_fwd(const float* in, float* out, size_t numElements, int neuron, float a, float b, void* stream);
/* Neuron::computeInputGrad (neuron.cuh:94-102): target = (add ? target : 0) + grad.  `acts` = neuron
 * output, `inputs` = neuron input (only read by abs / square).  target may alias actsGrad. */
int mcn_neuron_bwd(const float* actsGrad, const float* acts, const float* inputs, float* target,
                   size_t numElements, int neuron, float a, float b, int add, void* stream);

/* ---- NVMatrix verbs (nvmatrix.cuh:60-456) -------------------------------------------------- */
/* addProduct / rightMult (nvmatrix.cu:189-239, cublasSgemm).  Row-major
 * C[MxN] = beta*C + alpha*op(A)*op(B); transA: A is stored [KxM]; transB: B is stored [NxK]. */
size_t mcn_gemm_workspace_bytes(int M, int N, int K, int transA, int transB);
int mcn_gemm(const float* A, const float* B, float* C, int M, int N, int K, int transA, int transB,
             float alpha, float beta, void* workspace, size_t workspaceBytes, void* stream);
/* FCLayer::fpropActs (layer.cu:355-360) in one call: C = A*B (+ bias[row]) then neuron.  bias may be NULL. */
int mcn_gemm_bias_neuron(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                         float beta, int neuron, float neuronA, float neuronB,
                         void* workspace, size_t workspaceBytes, void* stream);
/* The same forward product for the layer directly under a softmax (FCLayer::fpropActs layer.cu:355-360 followed by
 * SoftmaxLayer::fpropActs layer.cu:585-595): additionally leaves stage 1 of the softmax -- per 64-row block and case,
 * max and sum of exp(x - max) -- in softmaxPart (mcn_softmax_workspace_bytes(M, N) bytes), for mcn_softmax_from_part.
 * Only shapes for which mcn_gemm_softmax_part_supported returns 1 (tall A, 16-byte aligned, N and K multiples of 4). */
int mcn_gemm_softmax_part_supported(const float* A, const float* B, const float* C, int M, int N, int K);
int mcn_gemm_bias_neuron_softmax_part(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                                      int neuron, float neuronA, float neuronB, void* softmaxPart, size_t softmaxPartBytes,
                                      void* stream);
/* FCLayer::fpropActs over ALL inputs of the layer in one launch (layer.cu:355-360 issues one addProduct per input; DeepID's
 * fc5 has two): acts (numOutputs, numCases) = sum_i weights[i] (numOutputs x numInputs[i], row-major) * inputs[i]
 * (numInputs[i], numCases) + biases[row], then the neuron.  The pointer arrays are host arrays of `count` entries.
 * Shapes: every numInputs[i] a multiple of 16, weights 16-byte aligned, count <= 4 (see ..._supported). */
int mcn_fc_fprop_multi_supported(const float* const* weights, const float* const* inputs, const int* numInputs, int count,
                                 const float* acts, int numOutputs, int numCases);
int mcn_fc_fprop_multi(const float* const* weights, const float* const* inputs, const int* numInputs, int count,
                       const float* biases, float* acts, int numOutputs, int numCases, int neuron, float neuronA, float neuronB,
                       void* stream);
/* FCLayer::bpropActs for ALL inputs in one launch (layer.cu:362-366): targets[i] (numInputs[i], numCases) =
 * scaleTargets[i]*targets[i] + g(weights[i]^T * actsGrad), where g is the identity or, when neuronActs[i] is given, the
 * gradient of the neuron that produced input i from its output neuronActs[i] (NeuronLayer::bpropActs layer.cu:245, for
 * neurons whose gradient needs their output only).  neuronActs / neuronTypes / neuronA / neuronB may be NULL. */
int mcn_fc_bprop_acts_multi_supported(const float* const* weights, float* const* targets, const int* numInputs, int count,
                                      const float* actsGrad, int numOutputs, int numCases);
int mcn_fc_bprop_acts_multi(const float* const* weights, float* const* targets, const int* numInputs, const float* scaleTargets,
                            const float* const* neuronActs, const int* neuronTypes, const float* neuronA, const float* neuronB,
                            int count, const float* actsGrad, int numOutputs, int numCases, void* stream);
/* FCLayer::bpropWeights over ALL inputs of the layer in one launch (layer.cu:372-382 issues one addProduct per input):
 * targets[i] (numOutputs x numInputs[i], row-major = the memory image of the reference's Fortran-order (inputs, outputs)
 * matrix) = scaleTargets[i]*targets[i] + scaleOutput[i] * actsGrad (numOutputs, numCases) * inputs[i]^T (numInputs[i], numCases).
 * The pointer / scale arrays are host arrays of `count` entries.  Shapes: numCases a multiple of 4, every numInputs[i]
 * even, operands 16-byte aligned, count <= 4 (see ..._supported). */
int mcn_fc_bprop_weights_multi_supported(const float* actsGrad, const float* const* inputs, float* const* targets,
                                         const int* numInputs, int count, int numOutputs, int numCases);
int mcn_fc_bprop_weights_multi(const float* actsGrad, const float* const* inputs, float* const* targets, const int* numInputs,
                               const float* scaleOutput, const float* scaleTargets, int count, int numOutputs, int numCases,
                               void* stream);
/* addVector (nvmatrix.cu:652-690): mat(rows x cols) += scale*vec; perRow=1: vec[rows], 0: vec[cols]. */
int mcn_add_vector(float* mat, const float* vec, int rows, int cols, int perRow, float scale, void* stream);
/* addSum (nvmatrix.cu:896-906 -> _aggregate :700-805): tgt = scaleThis*tgt + scaleSum*sum(mat, axis).
 * axis 1: row sums (tgt[rows]); axis 0: column sums (tgt[cols]). */
size_t mcn_add_sum_workspace_bytes(int rows, int cols, int axis);
int mcn_add_sum(float* tgt, const float* mat, int rows, int cols, int axis, float scaleThis, float scaleSum,
                void* workspace, size_t workspaceBytes, void* stream);
/* The same row sums (axis 1) for up to 8 matrices in one launch, plus one more launch when a matrix has rows longer than
 * 8192 columns: tgt_i = scaleThis_i*tgt_i + scaleSum_i*rowsums(mat_i), bit-identical to one mcn_add_sum per matrix.  The
 * reference issues an addSum per layer for its bias gradients (layer.cu:464-474, :545-549, FCLayer::bpropBiases); a
 * backward pass can hand all of them over at once. */
typedef struct mcn_row_sum_item {
    float* tgt;
    const float* mat;
    int rows, cols;
    float scaleThis, scaleSum;
} mcn_row_sum_item;
size_t mcn_add_sum_multi_workspace_bytes(const mcn_row_sum_item* items, int count);
int mcn_add_sum_multi(const mcn_row_sum_item* items, int count, void* workspace, size_t workspaceBytes, void* stream);
/* Only the first of the two launches: matrices whose rows have at most 8192 columns are finished; every other matrix i
 * leaves (rows_i x ceil(cols_i / 8192)) partial row sums in partialsOut[i] (mcn_add_sum_workspace_bytes(rows_i, cols_i, 1)
 * bytes), for mcn_sgd_plan_build_fused to finish inside the update. */
int mcn_add_sum_multi_stage1(const mcn_row_sum_item* items, int count, float* const* partialsOut, void* stream);
/* NVMatrix::add (nvmatrix.cuh eltwise WeightedAdd): a = sa*a + sb*b */
int mcn_axpby(float* a, const float* b, size_t n, float sa, float sb, void* stream);
/* NVMatrix::scale / copy with scale: out = s*in */
int mcn_scale(const float* in, float* out, size_t n, float s, void* stream);
/* eltwise max (EltwiseMaxLayer layer.cu:643-649): out = max(a, b) */
int mcn_eltwise_max(const float* a, const float* b, float* out, size_t n, void* stream);
/* computeEltwiseMaxGrad (layer_kernels.cuh, layer_kernels.cu:172-204) */
int mcn_eltwise_max_grad(const float* actGrad, const float* input, const float* output, float* target,
                         size_t n, int add, void* stream);
/* NVMatrix::sum() (nvmatrix.cu:936-979): *out (device double) = sum; accumulate != 0 adds to *out. */
int mcn_sum_all(const float* a, size_t n, double* outDevice, int accumulate, void* stream);
/* NVMatrix::transpose (hard transpose, worker.cu:201-208): dst (cols x rows) = src(rows x cols)^T */
int mcn_transpose(const float* src, float* dst, int rows, int cols, void* stream);

/* ---- image-op layers outside the DeepID stacks (conv_util.cuh:38-52) ------------------------------
 * All arrays (channels * pixels, numImages), images contiguous; rectangular maps where the reference is square-only.
 * convBedOfNails / convBedOfNailsUndo (conv_util.cu:665-680): sub-sampling at startX + k*strideX in both directions;
 * the undo zeroes the pixels between the nails when scaleTargets == 0 (the reference zeroes the matrix first) and leaves
 * them alone otherwise. */
int mcn_bed_of_nails(const float* images, float* target, int numImages, int numChannels, int imgSizeY, int imgSizeX, int startX,
                     int strideX, int outputsY, int outputsX, float scaleTargets, float scaleOutput, void* stream);
int mcn_bed_of_nails_undo(const float* actsGrad, float* target, int numImages, int numChannels, int imgSizeY, int imgSizeX,
                          int startX, int strideX, int outputsY, int outputsX, float scaleTargets, float scaleOutput, void* stream);
/* convGaussianBlur (conv_util.cuh:50, conv_util.cu:682-745): ONE direction of a separable blur with a symmetric filter of
 * filterSize (odd) taps in device memory, zero outside the image.  Not in place (the reference's register queue allows
 * target == images; here every output reads its neighbours from `images`). */
int mcn_gaussian_blur(const float* images, const float* filter, float* target, int numImages, int numChannels, int imgSizeY,
                      int imgSizeX, int filterSize, int horiz, float scaleTargets, float scaleOutput, void* stream);
/* convResizeBilinear (conv_util.cu:2416-2468): square maps, imgSize ~ scale * tgtSize */
int mcn_resize_bilinear(const float* images, float* target, int numImages, int numChannels, int imgSize, int tgtSize, float scale,
                        void* stream);
/* convRGBToYUV / convRGBToLAB (conv_util.cu:2470-2591): 3 channels in, 3 out; LAB raises to 0.3333 with the accurate powf
 * where the reference uses the fast intrinsic */
int mcn_rgb_to_yuv(const float* images, float* target, int numImages, int imgPixels, void* stream);
int mcn_rgb_to_lab(const float* images, float* target, int numImages, int imgPixels, int center, void* stream);
/* convShift (conv_util.cuh:52, conv_util.cu:2902-2990; on the HOST in the reference): a gather of pixels,
 * target[(f*numChannels + c)*tgtPixels + i, n] = images[c*imgPixels + int(filter[i*numFilters + f]), n];
 * filter (tgtPixels, numFilters) holds source pixel indices as floats, in device memory (indices are clamped to the image). */
int mcn_shift(const float* images, const float* filter, float* target, int numImages, int numChannels, int imgPixels, int tgtPixels,
              int numFilters, void* stream);

/* ---- layer_kernels.cuh:39-56 and SoftmaxLayer (layer.cu:585-595) ---------------------------- */
size_t mcn_softmax_workspace_bytes(int numOut, int numCases);
int mcn_softmax(const float* x, float* probs, int numOut, int numCases, void* workspace, size_t workspaceBytes,
                void* stream);
/* Stage 2 of mcn_softmax / mcn_softmax_logreg_grad alone, from the partials mcn_gemm_bias_neuron_softmax_part left
 * (labels and gradTarget both NULL: probabilities only). */
int mcn_softmax_from_part(const float* x, float* probs, const float* labels, float* gradTarget, float coeff, int numOut,
                          int numCases, const void* part, size_t partBytes, void* stream);
/* Softmax plus, in the same pass, what computeLogregSoftmaxGrad (layer_kernels.cu:287-308) would write for a logreg
 * cost on top of it: gradTarget[k,n] = coeff*((labels[n] == k) - probs[k,n]) (overwrite; bit-identical to the separate call). */
int mcn_softmax_logreg_grad(const float* x, float* probs, const float* labels, float* gradTarget, float coeff, int numOut,
                            int numCases, void* workspace, size_t workspaceBytes, void* stream);
/* computeLogregCost (layer_kernels.cuh:39, layer_kernels.cu:216-239) + the two NVMatrix::sum() of
 * LogregCostLayer::fpropActs (layer.cu:978-979): costOut[0] += -sum(logprob), costOut[1] += numCases - sum(correct)
 * (device doubles; pass NULL to skip). */
size_t mcn_logreg_cost_workspace_bytes(int numOut, int numCases);
int mcn_logreg_cost(const float* labels, const float* probs, float* labelLogProbs, float* correctProbs,
                    int numOut, int numCases, double* costOut, void* workspace, size_t workspaceBytes, void* stream);
/* SoftmaxLayer::fpropActs + LogregCostLayer::fpropActs (layer.cu:585-595, 969-981) without a second pass over the
 * probabilities: the softmax launch also writes, optionally, the logreg gradient through the softmax (gradTarget, may be
 * NULL), log p[label], and counts the entries tied at each column's maximum in integer atomics; a one-block launch then
 * finishes labelLogProbs / correctProbs / costOut (the values of mcn_logreg_cost).  finish != 0: that launch follows at
 * once; finish == 0: the caller issues mcn_softmax_logreg_cost_finish later (any stream ordered after this call), and
 * labelLogProbs / correctProbs / costOut are defined only after it.  part != NULL: stage 1 was left by
 * mcn_gemm_bias_neuron_softmax_part; else it runs first into workspace (mcn_softmax_workspace_bytes).  state:
 * mcn_softmax_logreg_cost_state_bytes(numCases) bytes of device memory, zero before the first call, used by one
 * call (+ its finish) at a time and left zero.  Needs numCases % 4 == 0 (mcn_softmax_logreg_cost_supported) and
 * 16-byte aligned matrices. */
size_t mcn_softmax_logreg_cost_state_bytes(int numCases);
int mcn_softmax_logreg_cost_supported(int numOut, int numCases);
int mcn_softmax_logreg_cost(const float* x, float* probs, const float* labels, float* gradTarget, float coeff, int numOut,
                            int numCases, const void* part, size_t partBytes, void* workspace, size_t workspaceBytes,
                            float* labelLogProbs, float* correctProbs, double* costOut, void* state, size_t stateBytes,
                            int finish, void* stream);
int mcn_softmax_logreg_cost_finish(float* labelLogProbs, float* correctProbs, double* costOut, void* state, size_t stateBytes,
                                   int numCases, void* stream);
/* computeLogregGrad (layer_kernels.cu:241-263) */
int mcn_logreg_grad(const float* labels, const float* probs, float* target, int numOut, int numCases, int add,
                    float coeff, void* stream);
/* computeSoftmaxGrad (layer_kernels.cu:265-285) */
int mcn_softmax_grad(const float* acts, const float* actsGrad, float* target, int numOut, int numCases, int add,
                     void* stream);
/* computeLogregSoftmaxGrad (layer_kernels.cu:287-308) */
int mcn_logreg_softmax_grad(const float* labels, const float* probs, float* target, int numOut, int numCases,
                            int add, float coeff, void* stream);
/* computeL2SNCost (layer_kernels.cu:505-604) incl. getOptimalThr (:482-498) run on the device.
 * output[numCases/2]; info (device, 4 floats, may be NULL): {m used, index of chosen pair or -1, 0, 0};
 * pairStats (device, 2*numCases/2 floats, may be NULL): dists then pair labels;
 * costOut (device double, may be NULL): += sum(output).
 * Multi-GPU: when extDists/extLabels (device, extCount entries each) are given the margin is searched
 * over those (the all-gathered pairs of every rank) instead of the local pairs. */
int mcn_l2sn_cost(const float* idens, const float* data, float* output, int numD, int numCases, float m,
                  float* info, float* pairStats, double* costOut,
                  const float* extDists, const float* extLabels, int extCount, void* stream);
/* computeL2SNGrad (layer_kernels.cu:620-735): dataTarget (numD, numCases) (+)= coeff * grad */
int mcn_l2sn_grad(const float* idens, const float* data, float* dataTarget, int numD, int numCases, float m,
                  int add, float coeff, const float* extDists, const float* extLabels, int extCount, void* stream);
/* The same gradient from what a preceding mcn_l2sn_cost on the same inputs left in device memory (pairStats and
 * info[0] = margin): one elementwise launch, so that the layer's backward pass does not repeat the statistics and
 * the margin search (layer.cu:1073-1082 calls computeL2SNGrad with the margin its fprop kept). */
int mcn_l2sn_grad_from_stats(const float* data, float* dataTarget, int numD, int numCases, const float* pairStats,
                             const float* info, int add, float coeff, void* stream);
/* FCLayer::bpropActs (layer.cu:368-373) of a classifier layer + the L2-SN gradient of the same input matrix
 * (L2SNCostLayer::bpropActs layer.cu:1073-1082) in the product's own reduce pass: targets (numInputs, numCases) =
 * weights^T actsGrad + coeff * dL2SN/d(l2snData), bitwise what mcn_gemm(transA = 1) followed by
 * mcn_l2sn_grad_from_stats(add = 1) leaves.  weights (numOutputs, numInputs), actsGrad (numOutputs, numCases); pairStats / info
 * as left by mcn_l2sn_pair_stats / mcn_l2sn_cost_from_stats, possibly on another stream: then gate / gateValue / timeoutMs name
 * the stream gate (mcn_stream_signal) that announces them and the kernel waits for it itself (gate = NULL: already ordered).
 * Shapes of the K-split form only (mcn_fc_bprop_acts_l2sn_supported = 1); workspace as mcn_gemm_workspace_bytes(.., 1, 0). */
int mcn_fc_bprop_acts_l2sn_supported(const float* weights, const float* actsGrad, const float* targets, int numInputs,
                                     int numCases, int numOutputs);
int mcn_fc_bprop_acts_l2sn(const float* weights, const float* actsGrad, float* targets, int numInputs, int numCases,
                           int numOutputs, void* workspace, size_t workspaceBytes, const float* l2snData,
                           const float* pairStats, const float* info, float coeff, int* gate, int gateValue, int timeoutMs,
                           void* stream);
/* Pair distances / labels only (first half of computeL2SNCost, :517-561): pairStats[2*numCases/2].  Several workgroups, one per
 * window of 8 pairs; every distance has the bits mcn_l2sn_cost computes. */
int mcn_l2sn_pair_stats(const float* idens, const float* data, float* pairStats, int numD, int numCases, void* stream);
/* Second half of computeL2SNCost (:563-604, margin search :482-498) from the pair statistics mcn_l2sn_pair_stats left in device
 * memory: mcn_l2sn_pair_stats + mcn_l2sn_cost_from_stats == mcn_l2sn_cost with pairStats, bit for bit.  Arguments as there. */
int mcn_l2sn_cost_from_stats(const float* pairStats, float* output, int numCases, float m, float* info, double* costOut,
                             const float* extDists, const float* extLabels, int extCount, void* stream);
/* computeFisher2Cost / computeFisher2Grad (layer_kernels.cu:310-369, :371-479; kLabel2Weight :32-47).
 * costOut (device double, may be NULL) += cost * numCases (Fisher2CostLayer::fpropActs layer.cu:1025-1036);
 * dataTarget (numD, numCases) (+)= coeff * gradient.  workspace: mcn_fisher2_workspace_bytes(numCases). */
size_t mcn_fisher2_workspace_bytes(int numCases);
int mcn_fisher2_cost(const float* labels, const float* data, int numD, int numCases, float alpha, double* costOut,
                     void* workspace, size_t workspaceBytes, void* stream);
int mcn_fisher2_grad(const float* labels, const float* data, float* dataTarget, int numD, int numCases, float alpha, int add,
                     float coeff, void* workspace, size_t workspaceBytes, void* stream);
/* computeL2regCost / computeL2regGrad (layer_kernels.cu:744-794) */
int mcn_l2reg_cost(const float* ground, const float* data, float* output, int numD, int numCases,
                   double* costOut, void* stream);
int mcn_l2reg_grad(const float* ground, const float* data, float* target, int numD, int numCases, int add,
                   float coeff, void* stream);

/* ---- weights.cuh:150-163 --------------------------------------------------------------------- */
/* Weights::update in one pass: useGrad: inc = mom*inc + grad; wc>0: inc -= wc*eps*W; W += inc. */
int mcn_sgd_update(float* W, float* inc, const float* grad, size_t n, float mom, float wc, float eps,
                   int useGrad, void* stream);

/* The same update for every matrix of a net in ONE launch (the reference launches up to three
 * kEltwiseBinaryOp per matrix, weights.cuh:150-163).  Build the plan once (uploads `count` records to
 * devicePlan, blocking), then launch it every step.  All buffers must be 16-byte aligned. */
typedef struct mcn_sgd_item {
    float* W;
    float* inc;
    const float* grad; /* read only when useGrad != 0 */
    size_t n;
    float mom, wc, eps;
    int useGrad;
} mcn_sgd_item;
size_t mcn_sgd_plan_bytes(int count);
int mcn_sgd_plan_build(const mcn_sgd_item* items, int count, void* devicePlan, size_t planBytes, int* numBlocksOut, void* stream);
int mcn_sgd_update_planned(const void* devicePlan, int count, int numBlocks, void* stream);
/* A plan whose update also FINISHES gradients that are still in pieces when it starts (no counterpart in the reference,
 * which runs addSum over its partialSum temporary as a launch of its own, layer.cu:486-495): partials[i].data == NULL
 * for an ordinary item; kind 1: `count` slabs of items[i].n floats whose sum is the gradient (the module-chunk partials of
 * mcn_conv_weight_acts_partials); kind 2: items[i].n rows of `count` partial row sums, gradient = scale * row sum (stage 1
 * of mcn_add_sum_multi_stage1).  The sums use the code and order of the stand-alone second stages, so the updated weights
 * are bit-identical to reduce-then-update; the finished gradient is also stored to items[i].grad. */
typedef struct mcn_sgd_partials {
    const float* data;
    int count, kind;
    float scale;
} mcn_sgd_partials;
int mcn_sgd_plan_build_fused(const mcn_sgd_item* items, const mcn_sgd_partials* partials, int count, void* devicePlan,
                             size_t planBytes, int* numBlocksOut, void* stream);

/* ---- data-parallel exchange (no counterpart in the reference, which is single-GPU: convnet.py:52-53).
 * RCCL is loaded lazily; one communicator per process (one process per GPU).
 * TEST BUILD ONLY (libmiconvnet_hooks.so, -DMCN_TEST_HOOKS; the product library refuses it): with MCN_COMM=shm in the
 * environment of the rank that calls mcn_comm_unique_id, the id names a POSIX shared-memory segment instead and the
 * communicator stages every collective through it (fixed rank-order sums, queued on the stream like RCCL's): a rehearsal
 * transport for ranks that share ONE GPU, where RCCL refuses the second rank.  Same entry points, same results on every rank. */
int mcn_comm_unique_id(char idBytes[128]);
int mcn_comm_init(void** comm, int numRanks, int rank, const char idBytes[128]);
int mcn_comm_destroy(void* comm);
/* the ranks the communicator itself counts (ncclCommCount), RCCL's version code (ncclGetVersion), isRccl = 0 for the
 * rehearsal transport */
int mcn_comm_info(void* comm, int* numRanks, int* version, int* isRccl);
int mcn_allreduce_sum(void* comm, float* buf, size_t count, void* stream);
int mcn_allgather(void* comm, const float* send, float* recv, size_t countPerRank, void* stream);
/* collectives issued between begin and end are fused into one launch (ncclGroupStart / ncclGroupEnd) */
int mcn_comm_group_begin(void);
int mcn_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_CONVNET_H */
