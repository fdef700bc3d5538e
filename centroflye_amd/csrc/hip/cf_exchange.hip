// cf_exchange.hip — the multi-GPU steps of the recruit + distance path, inside the library, on the
// context's stream, over cf_comm (RCCL in the product build).  One process per GPU, reads sharded across ranks.
//
// The reference has no distributed code (every collective is new design).  Per step and rank:
//   cf_count_kmers        A1 on the local read shard                                           (no traffic)
//   cf_exchange_table     local (key, pres, multi) triples bucketed by owner = hash(key) % n on the device, ONE
//                         all-to-all of 16-byte records (send/recv pairs on all xGMI links at once), the owner adds
//                         what it receives into a fresh table -> exact global counts of the owned keys
//   cf_select_rare        A2 on the owned keys
//   cf_allgather_kmers    all-gather of the (small) rare lists; every rank sorts and installs the union
//   cf_build_clouds       A3 on the local shard
//   cf_allgather_clouds   all-gather of the per-unit clouds (CSR) of every rank: the distance stage's view
//   cf_dist_edges         A5 + A6 on first k-mers a with a % n == rank over ALL clouds — no reduction
//   cf_allreduce_unique   OR of the selected-k-mer masks
#include "cf_comm.h"

void cf_free_table(cf_ctx* c);
void cf_free_kmers(cf_ctx* c);
void cf_free_gview(cf_ctx* c);
int cf_install_kmers(cf_ctx* ctx, int32_t k);          // cf_clouds.hip
int cf_refresh_unique_count(cf_ctx* ctx);               // cf_clouds.hip
int cf_table_ensure(cf_ctx* ctx, uint64_t want_cap);    // cf_count.hip

struct alignas(16) cf_pair { unsigned long long key, val; };

// owner of a key: bits of the mixer the table index (low bits) does not use
__device__ __forceinline__ uint32_t cf_owner_of(unsigned long long key, uint32_t world) {
    return (uint32_t)((cf_mix64(key ^ 0x5bf03635ull) >> 33) % world);
}

#define XCH_THREADS 256

// Table scan.  mode 0: per-workgroup counts of occupied slots per owner -> block_cnt[owner * gridDim + block];
// mode 1: block_cnt holds the exclusive scan of those counts (owner-major: every owner's records are contiguous);
// records are written at block_cnt[..] + LDS cursor, ranked inside the wave per distinct owner (no global atomics).
__global__ void __launch_bounds__(XCH_THREADS)
cf_xch_bucket_kernel(const cf_slot* __restrict__ table, uint64_t cap, uint32_t world, int mode, int64_t* __restrict__ block_cnt,
                     cf_pair* __restrict__ out) {
    uint32_t* cur = (uint32_t*)cf_lds;      // world counters
    const int lane = threadIdx.x & 63;
    for (uint32_t w = threadIdx.x; w < world; w += XCH_THREADS) cur[w] = 0;
    __syncthreads();
    const uint64_t chunk = ((cap + gridDim.x - 1) / gridDim.x + XCH_THREADS - 1) / XCH_THREADS * XCH_THREADS;
    const uint64_t c0 = (uint64_t)blockIdx.x * chunk, c1 = min(cap, c0 + chunk);
    for (uint64_t i0 = c0; i0 < c1; i0 += XCH_THREADS) {
        const uint64_t i = i0 + threadIdx.x;
        cf_slot sl; sl.key = 0; sl.val = 0;
        if (i < c1) sl = table[i];
        const bool occ = sl.key != 0ull;
        const unsigned long long key = sl.key & ~CF_OCC;
        const uint32_t own = occ ? cf_owner_of(key, world) : 0u;
        unsigned long long todo = __ballot(occ);
        while (todo) {                                  // one round per distinct owner present in the wave
            const int first = __ffsll((long long)todo) - 1;
            const uint32_t o = (uint32_t)__shfl((int)own, first);
            const unsigned long long m = __ballot(occ && own == o);
            uint32_t base = 0;
            if (lane == first) base = atomicAdd(&cur[o], (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, first);
            if (mode == 1 && occ && own == o) {
                const int64_t pos = block_cnt[(int64_t)o * gridDim.x + blockIdx.x] + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                out[pos] = cf_pair{key, sl.val};
            }
            todo &= ~m;
        }
    }
    if (mode == 0) {
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < world; w += XCH_THREADS) block_cnt[(int64_t)w * gridDim.x + blockIdx.x] = (int64_t)cur[w];
    }
}

__device__ __forceinline__ void cf_xch_table_add(cf_slot* __restrict__ table, uint64_t mask, unsigned long long key, unsigned long long inc,
                                                 unsigned int* __restrict__ flags) {
    const unsigned long long want = key | CF_OCC;
    uint64_t h = cf_mix64(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        unsigned long long cur = table[h].key;
        if (cur == 0ull) cur = atomicCAS(&table[h].key, 0ull, want);
        if (cur == 0ull || cur == want) { atomicAdd(&table[h].val, inc); return; }
        h = (h + 1) & mask;
    }
    atomicOr(flags, 1u);
}

__global__ void __launch_bounds__(256)
cf_xch_merge_kernel(cf_slot* __restrict__ table, uint64_t tmask, const cf_pair* __restrict__ recs, int64_t n, unsigned int* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const cf_pair r = recs[i];
        cf_xch_table_add(table, tmask, r.key, r.val, flags);
    }
}

__global__ void __launch_bounds__(256)
cf_xch_diff_kernel(const int64_t* __restrict__ ptr, int64_t n, int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = ptr[i + 1] - ptr[i];
}

// out[i] = bytes[i] | bits (bit i)  /  bits |= bytes : the unique mask travels as bytes (ncclMax on u8 = OR)
__global__ void __launch_bounds__(256)
cf_xch_bits_to_bytes_kernel(const uint32_t* __restrict__ bits, int64_t n, uint8_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint8_t)((bits[i >> 5] >> (i & 31)) & 1u);
}
__global__ void __launch_bounds__(256)
cf_xch_bytes_to_bits_kernel(uint32_t* __restrict__ bits, int64_t n, const uint8_t* __restrict__ in) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (in[i] && !((bits[i >> 5] >> (i & 31)) & 1u)) atomicOr(&bits[i >> 5], 1u << (i & 31));
}

void cf_free_gview(cf_ctx* c) {
    cf_release_t(c, c->g_unit_ptr, (size_t)c->g_reads + 1);
    cf_release_t(c, c->g_cloud_ptr, (size_t)c->g_units + 1);
    cf_release_t(c, c->g_entries_d, (size_t)c->g_entries);
    c->g_reads = c->g_units = c->g_entries = 0;
    c->g_h_unit_ptr.clear();
    c->have_gview = false;
}

namespace {

// temporaries of one call: released in reverse order when the guard goes out of scope
struct Bufs {
    cf_ctx* ctx;
    std::vector<std::pair<void*, size_t>> v;
    explicit Bufs(cf_ctx* c) : ctx(c) {}
    ~Bufs() { for (auto it = v.rbegin(); it != v.rend(); ++it) cf_release(ctx, it->first, it->second); }
    template <class T> int get(T** p, size_t n, const char* what) {
        int rc = cf_alloc_t(ctx, p, n, what);
        if (rc == 0) v.emplace_back((void*)*p, n * sizeof(T));
        return rc;
    }
    void keep(void* p) { for (auto& e : v) if (e.first == p) e.first = nullptr; }   // cf_release(nullptr) is a no-op
};

int comm_fail(cf_ctx* ctx, int rc, const std::string& err) { return cf_fail(ctx, rc, err); }

// every rank's n (one int64 each)
int gather_counts(cf_ctx* ctx, int64_t mine, std::vector<int64_t>& all) {
    cf_comm* cm = ctx->comm;
    Bufs tmp(ctx);
    int64_t *d_one = nullptr, *d_all = nullptr;
    CF_TRY(tmp.get(&d_one, 1, "comm count"));
    CF_TRY(tmp.get(&d_all, (size_t)cm->world, "comm counts"));
    CF_HIP(hipMemcpyAsync(d_one, &mine, 8, hipMemcpyHostToDevice, ctx->stream));
    CF_HIP(hipStreamSynchronize(ctx->stream));
    std::string err;
    int rc = cm->allgather(d_one, d_all, 8, ctx->stream, err);
    if (rc) return comm_fail(ctx, rc, err);
    all.assign((size_t)cm->world, 0);
    CF_HIP(hipMemcpy(all.data(), d_all, (size_t)cm->world * 8, hipMemcpyDeviceToHost));
    return 0;
}

// all-gather of device arrays of different lengths (bytes): every rank sends its array to every peer (send/recv pairs:
// all links at once); *d_out (allocated here, sum of counts) holds them in rank order
int allgatherv(cf_ctx* ctx, Bufs& keep, const void* d_mine, int64_t my_bytes, char** d_out, std::vector<int64_t>& bytes, const char* what) {
    cf_comm* cm = ctx->comm;
    CF_TRY(gather_counts(ctx, my_bytes, bytes));
    const int W = cm->world;
    std::vector<int64_t> soff((size_t)W, 0), sb((size_t)W, my_bytes), roff((size_t)W, 0);
    int64_t tot = 0;
    for (int p = 0; p < W; ++p) { roff[(size_t)p] = tot; tot += bytes[(size_t)p]; }
    CF_TRY(keep.get(d_out, (size_t)tot, what));
    std::string err;
    int rc = cm->alltoallv(d_mine, soff.data(), sb.data(), *d_out, roff.data(), bytes.data(), ctx->stream, err);
    if (rc) return comm_fail(ctx, rc, err);
    return 0;
}

}  // namespace

void cf_comm_apply_params(cf_ctx* ctx) {
    if (!ctx->comm) return;
    ctx->comm->round_bytes = ctx->comm_round_bytes;
    ctx->comm->self_p2p = ctx->comm_self_p2p != 0;
}

extern "C" {

int cf_comm_init(cf_ctx* ctx, int32_t rank, int32_t world, const char* rendezvous) {
    if (!ctx) return -22;
    if (ctx->comm) return cf_fail(ctx, -22, "cf_comm_init: already initialised");
    CF_HIP(hipSetDevice(ctx->device));
    std::string err;
    cf_comm* c = cf_comm_open(ctx->device, rank, world, rendezvous, err);
    if (!c) return cf_fail(ctx, -5, err);
    ctx->comm = c;
    cf_comm_apply_params(ctx);
    return 0;
}

int cf_comm_free(cf_ctx* ctx) {
    if (!ctx) return -22;
    if (ctx->comm) { (void)hipSetDevice(ctx->device); delete ctx->comm; ctx->comm = nullptr; }
    return 0;
}

int cf_comm_info(cf_ctx* ctx, int32_t* rank, int32_t* world) {
    if (!ctx) return -22;
    if (rank) *rank = ctx->comm ? ctx->comm->rank : 0;
    if (world) *world = ctx->comm ? ctx->comm->world : 1;
    return 0;
}

// vals[n] (host) := sum / max over ranks; op 0 = sum, 1 = max.  Also the barrier of the bench harness.
int cf_comm_allreduce_i64(cf_ctx* ctx, int64_t* vals, int64_t n, int32_t op) {
    if (!ctx || !vals || n < 0) return -22;
    if (!ctx->comm) return cf_fail(ctx, -22, "cf_comm_allreduce_i64: no communicator (cf_comm_init)");
    if (n == 0) return 0;
    CF_HIP(hipSetDevice(ctx->device));
    Bufs tmp(ctx);
    int64_t* d = nullptr;
    CF_TRY(tmp.get(&d, (size_t)n, "allreduce values"));
    CF_HIP(hipMemcpyAsync(d, vals, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    CF_HIP(hipStreamSynchronize(ctx->stream));
    std::string err;
    int rc = ctx->comm->allreduce(d, n, CF_COMM_I64, op == 1 ? CF_COMM_MAX : CF_COMM_SUM, ctx->stream, err);
    if (rc) return cf_fail(ctx, rc, err);
    CF_HIP(hipMemcpy(vals, d, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}

int cf_exchange_table(cf_ctx* ctx, int64_t* bytes_sent) {
    if (!ctx) return -22;
    if (!ctx->comm) return cf_fail(ctx, -22, "cf_exchange_table: no communicator (cf_comm_init)");
    if (!ctx->d_table) return cf_fail(ctx, -22, "cf_exchange_table: no table (call cf_count_kmers first)");
    CF_HIP(hipSetDevice(ctx->device));
    cf_comm* cm = ctx->comm;
    const uint32_t W = (uint32_t)cm->world;
    const int grid = std::max(1, ctx->n_cu) * 8;
    Bufs tmp(ctx);
    int64_t* d_blk = nullptr;
    CF_TRY(tmp.get(&d_blk, (size_t)W * grid + 1, "exchange block counts"));
    const size_t lds = (size_t)W * 4 + 16;
    // 1. owner histogram per workgroup, scan (owner-major), scatter into the send buffer
    hipLaunchKernelGGL(cf_xch_bucket_kernel, dim3((unsigned)grid), dim3(XCH_THREADS), lds, ctx->stream, (const cf_slot*)ctx->d_table,
                       (uint64_t)ctx->table_cap, W, 0, d_blk, (cf_pair*)nullptr);
    CF_KERNEL_CHECK("cf_xch_bucket_kernel");
    CF_HIP(hipMemsetAsync(d_blk + (size_t)W * grid, 0, 8, ctx->stream));
    int64_t n_local = 0;
    CF_TRY(cf_scan_exclusive_i64(ctx, d_blk, d_blk, (int64_t)W * grid + 1, &n_local));
    std::vector<int64_t> start((size_t)W + 1, 0);
    for (uint32_t w = 0; w < W; ++w) CF_HIP(hipMemcpy(&start[w], d_blk + (size_t)w * grid, 8, hipMemcpyDeviceToHost));
    start[W] = n_local;
    cf_pair* d_send = nullptr;
    CF_TRY(tmp.get(&d_send, (size_t)n_local, "exchange send records"));
    if (n_local) {
        hipLaunchKernelGGL(cf_xch_bucket_kernel, dim3((unsigned)grid), dim3(XCH_THREADS), lds, ctx->stream, (const cf_slot*)ctx->d_table,
                           (uint64_t)ctx->table_cap, W, 1, d_blk, d_send);
        CF_KERNEL_CHECK("cf_xch_bucket_kernel");
    }
    CF_HIP(hipStreamSynchronize(ctx->stream));
    // 2. who sends how much to whom: all-gather of the W send counts of every rank
    std::vector<int64_t> sb((size_t)W), soff((size_t)W), rb((size_t)W), roff((size_t)W);
    for (uint32_t w = 0; w < W; ++w) { soff[w] = start[w] * (int64_t)sizeof(cf_pair); sb[w] = (start[w + 1] - start[w]) * (int64_t)sizeof(cf_pair); }
    {
        int64_t *d_sc = nullptr, *d_all = nullptr;
        CF_TRY(tmp.get(&d_sc, (size_t)W, "exchange send counts"));
        CF_TRY(tmp.get(&d_all, (size_t)W * W, "exchange count matrix"));
        CF_HIP(hipMemcpy(d_sc, sb.data(), (size_t)W * 8, hipMemcpyHostToDevice));
        std::string err;
        int rc = cm->allgather(d_sc, d_all, (int64_t)W * 8, ctx->stream, err);
        if (rc) return cf_fail(ctx, rc, err);
        std::vector<int64_t> all((size_t)W * W);
        CF_HIP(hipMemcpy(all.data(), d_all, (size_t)W * W * 8, hipMemcpyDeviceToHost));
        for (uint32_t p = 0; p < W; ++p) rb[p] = all[(size_t)p * W + (size_t)cm->rank];
    }
    int64_t n_recv_bytes = 0;
    for (uint32_t p = 0; p < W; ++p) { roff[p] = n_recv_bytes; n_recv_bytes += rb[p]; }
    const int64_t n_recv = n_recv_bytes / (int64_t)sizeof(cf_pair);
    cf_pair* d_recv = nullptr;
    CF_TRY(tmp.get(&d_recv, (size_t)n_recv, "exchange receive records"));
    // 3. the all-to-all
    {
        std::string err;
        int rc = cm->alltoallv(d_send, soff.data(), sb.data(), d_recv, roff.data(), rb.data(), ctx->stream, err);
        if (rc) return cf_fail(ctx, rc, err);
    }
    // 4. the owner's table: fresh, sized for what arrived (an upper bound of the distinct owned keys), records added
    const uint64_t cap = cf_pow2_ceil((uint64_t)std::max<int64_t>(2 * n_recv, 1024));
    CF_TRY(cf_table_ensure(ctx, cap));
    CF_HIP(hipMemsetAsync(ctx->d_table, 0, (size_t)cap * sizeof(cf_slot), ctx->stream));
    unsigned int* d_flags = nullptr;
    CF_TRY(tmp.get(&d_flags, 4, "exchange flags"));
    CF_HIP(hipMemsetAsync(d_flags, 0, 16, ctx->stream));
    if (n_recv) {
        hipLaunchKernelGGL(cf_xch_merge_kernel, dim3((unsigned)cf_grid_for(n_recv, 256, grid)), dim3(256), 0, ctx->stream, ctx->d_table,
                           (uint64_t)(cap - 1), (const cf_pair*)d_recv, n_recv, d_flags);
        CF_KERNEL_CHECK("cf_xch_merge_kernel");
    }
    unsigned int flags = 0;
    CF_HIP(hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost));
    if (flags & 1u) return cf_fail(ctx, -34, "cf_exchange_table: owner table full");
    ctx->exchange_bytes = n_local * (int64_t)sizeof(cf_pair) - sb[(size_t)cm->rank];
    if (bytes_sent) *bytes_sent = ctx->exchange_bytes;
    return 0;
}

int cf_allgather_kmers(cf_ctx* ctx, int64_t* n_out) {
    if (!ctx) return -22;
    if (!ctx->comm) return cf_fail(ctx, -22, "cf_allgather_kmers: no communicator (cf_comm_init)");
    if (!ctx->d_lut) return cf_fail(ctx, -22, "cf_allgather_kmers: no k-mer set selected");
    CF_HIP(hipSetDevice(ctx->device));
    Bufs tmp(ctx);
    char* d_all = nullptr;
    std::vector<int64_t> bytes;
    CF_TRY(allgatherv(ctx, tmp, ctx->d_kmers, ctx->n_kmers * 8, &d_all, bytes, "gathered k-mers"));
    int64_t n = 0;
    for (int64_t b : bytes) n += b / 8;
    if (n >= (int64_t)1 << 31) return cf_fail(ctx, -34, "more than 2^31 selected k-mers");
    unsigned long long* d_tmp = nullptr;
    CF_TRY(tmp.get(&d_tmp, (size_t)n, "sort scratch"));
    const int k = ctx->set_k;
    CF_TRY(cf_radix_sort_u64(ctx, (unsigned long long*)d_all, d_tmp, n, 2 * k));    // owners hold disjoint keys: the union is unique
    cf_free_kmers(ctx);
    tmp.keep(d_all);
    ctx->d_kmers = (unsigned long long*)d_all;
    ctx->n_kmers = n;
    // the block was allocated with n * 8 bytes: matches cf_free_kmers' accounting
    CF_TRY(cf_install_kmers(ctx, k));
    if (n_out) *n_out = n;
    return 0;
}

int cf_allgather_clouds(cf_ctx* ctx, int64_t* n_entries) {
    if (!ctx) return -22;
    if (!ctx->comm) return cf_fail(ctx, -22, "cf_allgather_clouds: no communicator (cf_comm_init)");
    if (!ctx->have_clouds) return cf_fail(ctx, -22, "cf_allgather_clouds: no clouds built");
    CF_HIP(hipSetDevice(ctx->device));
    cf_free_gview(ctx);
    const int64_t R = ctx->n_reads, U = ctx->n_units, N = ctx->n_entries;
    const int grid = std::max(1, ctx->n_cu) * 8;
    Bufs tmp(ctx);
    int64_t *d_upr = nullptr, *d_sizes = nullptr;
    CF_TRY(tmp.get(&d_upr, (size_t)R + 1, "units per read"));
    CF_TRY(tmp.get(&d_sizes, (size_t)U + 1, "cloud sizes"));
    if (R) hipLaunchKernelGGL(cf_xch_diff_kernel, dim3((unsigned)cf_grid_for(R, 256, grid)), dim3(256), 0, ctx->stream, (const int64_t*)ctx->d_unit_ptr, R, d_upr);
    if (U) hipLaunchKernelGGL(cf_xch_diff_kernel, dim3((unsigned)cf_grid_for(U, 256, grid)), dim3(256), 0, ctx->stream, (const int64_t*)ctx->d_cloud_ptr, U, d_sizes);
    CF_KERNEL_CHECK("cf_xch_diff_kernel");
    CF_HIP(hipStreamSynchronize(ctx->stream));
    char *g_upr = nullptr, *g_sizes = nullptr;
    std::vector<int64_t> b_upr, b_sizes;
    CF_TRY(allgatherv(ctx, tmp, d_upr, R * 8, &g_upr, b_upr, "gathered units per read"));
    CF_TRY(allgatherv(ctx, tmp, d_sizes, U * 8, &g_sizes, b_sizes, "gathered cloud sizes"));
    int64_t Rg = 0, Ug = 0;
    for (int64_t b : b_upr) Rg += b / 8;
    for (int64_t b : b_sizes) Ug += b / 8;
    if (Ug >= (int64_t)1 << 31) return cf_fail(ctx, -34, "more than 2^31 units over all ranks");
    // entries last: the big one (4 N_ce bytes over all ranks)
    {
        std::vector<int64_t> cnt;     // (allocated directly: the view owns it)
        CF_TRY(gather_counts(ctx, N * 4, cnt));
        int64_t tot = 0;
        std::vector<int64_t> soff(cnt.size(), 0), sb(cnt.size(), N * 4), roff(cnt.size(), 0);
        for (size_t p = 0; p < cnt.size(); ++p) { roff[p] = tot; tot += cnt[p]; }
        ctx->g_entries = tot / 4;
        CF_TRY(cf_alloc_t(ctx, &ctx->g_entries_d, (size_t)ctx->g_entries, "gathered cloud entries"));
        std::string err;
        int rc = ctx->comm->alltoallv(ctx->d_entries, soff.data(), sb.data(), ctx->g_entries_d, roff.data(), cnt.data(), ctx->stream, err);
        if (rc) { cf_free_gview(ctx); return cf_fail(ctx, rc, err); }
    }
    ctx->g_reads = Rg; ctx->g_units = Ug;
    int rc = cf_alloc_t(ctx, &ctx->g_unit_ptr, (size_t)Rg + 1, "gathered unit_ptr");
    if (rc == 0) rc = cf_alloc_t(ctx, &ctx->g_cloud_ptr, (size_t)Ug + 1, "gathered cloud_ptr");
    int64_t tot_u = 0, tot_e = 0;
    // exclusive scans over n + 1 elements (the last input is unused): ptr[n] = total
    if (rc == 0) {
        int64_t *s_upr = nullptr, *s_sizes = nullptr;
        rc = tmp.get(&s_upr, (size_t)Rg + 1, "scan input");
        if (rc == 0) rc = tmp.get(&s_sizes, (size_t)Ug + 1, "scan input");
        if (rc == 0) {
            hipError_t e = hipMemsetAsync(s_upr, 0, (size_t)(Rg + 1) * 8, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(s_sizes, 0, (size_t)(Ug + 1) * 8, ctx->stream);
            if (e == hipSuccess && Rg) e = hipMemcpyAsync(s_upr, g_upr, (size_t)Rg * 8, hipMemcpyDeviceToDevice, ctx->stream);
            if (e == hipSuccess && Ug) e = hipMemcpyAsync(s_sizes, g_sizes, (size_t)Ug * 8, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) rc = cf_fail(ctx, -5, std::string("cf_allgather_clouds: ") + hipGetErrorString(e));
        }
        if (rc == 0) rc = cf_scan_exclusive_i64(ctx, s_upr, ctx->g_unit_ptr, Rg + 1, &tot_u);
        if (rc == 0) rc = cf_scan_exclusive_i64(ctx, s_sizes, ctx->g_cloud_ptr, Ug + 1, &tot_e);
    }
    if (rc == 0 && (tot_u != Ug || tot_e != ctx->g_entries)) rc = cf_fail(ctx, -5, "cf_allgather_clouds: gathered sizes do not add up");
    if (rc == 0) {
        ctx->g_h_unit_ptr.assign((size_t)Rg + 1, 0);
        if (hipMemcpy(ctx->g_h_unit_ptr.data(), ctx->g_unit_ptr, (size_t)(Rg + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = cf_fail(ctx, -5, "cf_allgather_clouds: unit_ptr copy");
    }
    if (rc) { cf_free_gview(ctx); return rc; }
    ctx->have_gview = true;
    if (n_entries) *n_entries = ctx->g_entries;
    return 0;
}

int cf_allreduce_unique(cf_ctx* ctx, int64_t* n_unique) {
    if (!ctx) return -22;
    if (!ctx->comm) return cf_fail(ctx, -22, "cf_allreduce_unique: no communicator (cf_comm_init)");
    if (!ctx->d_unique_bits) return cf_fail(ctx, -22, "cf_allreduce_unique: no k-mer set");
    CF_HIP(hipSetDevice(ctx->device));
    const int64_t n = ctx->n_kmers;
    if (n) {
        Bufs tmp(ctx);
        uint8_t* d_bytes = nullptr;
        CF_TRY(tmp.get(&d_bytes, (size_t)n, "unique mask bytes"));
        const int grid = cf_grid_for(n, 256, std::max(1, ctx->n_cu) * 8);
        hipLaunchKernelGGL(cf_xch_bits_to_bytes_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->d_unique_bits, n, d_bytes);
        CF_KERNEL_CHECK("cf_xch_bits_to_bytes_kernel");
        CF_HIP(hipStreamSynchronize(ctx->stream));
        std::string err;
        int rc = ctx->comm->allreduce(d_bytes, n, CF_COMM_U8, CF_COMM_MAX, ctx->stream, err);
        if (rc) return cf_fail(ctx, rc, err);
        hipLaunchKernelGGL(cf_xch_bytes_to_bits_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ctx->d_unique_bits, n, (const uint8_t*)d_bytes);
        CF_KERNEL_CHECK("cf_xch_bytes_to_bits_kernel");
        CF_HIP(hipStreamSynchronize(ctx->stream));
    }
    CF_TRY(cf_refresh_unique_count(ctx));
    if (n_unique) *n_unique = ctx->stats.n_unique;
    return 0;
}

}  // extern "C"
