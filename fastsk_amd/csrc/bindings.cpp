// bindings.cpp — pybind11 module `fastsk_amd._fastsk`: the reference's Python surface
// (QData/FastSK src/fastsk/_fastsk/bindings.cpp:12-51) on top of the C ABI in
// include/fastsk_amd.h. Same class name, constructor keywords and defaults, method names and
// return types; host C++ only — all computation happens behind fsk_* in libfastsk_amd.so.
//
// Deviations, all documented in INTEGRATION.md:
//   - errors raise ValueError / RuntimeError instead of printf + exit(1) (fastsk.cpp:53-58);
//   - fit() / score() (LIBSVM, fastsk.cpp:239-530) are outside this path and raise
//     NotImplementedError (they are unusable from Python in the reference as well);
//   - additive keyword arguments (device, devices, collective, deadline_ms, path, seed, skip_test_block), numpy and
//     DLPack getters. devices=[0,1,...]: one engine per listed GPU behind the same object (fsk_create_multi) —
//     the reference parallelises the same call over t host threads (fastsk_kernel.cpp:54-93).
//   - skip_test_block (default False: compute_kernel computes the whole N x N triangle, as fastsk.cpp:30-118 does).
//     The test x test block, which no getter of the reference exposes (fastsk.cpp:190-217), can be left out:
//     skip_test_block=True never computes it (its cells read as zero); skip_test_block="lazy" (or None) leaves it
//     out of compute_kernel and the first request that needs it (get_block over test x test cells, get_counts_np,
//     counts_digest over test rows, get_triangle_dlpack, save_kernel) runs the whole compute once more with
//     nothing left out — about twice the time of compute_kernel in total; the object keeps the tokens until then.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fastsk_amd.h"

namespace py = pybind11;

namespace {

struct Flat {
    std::vector<int32_t> tokens;
    std::vector<int64_t> offsets{0};
    // The rows of X — what FastaUtility.read_data returns (a list of lists of ints), or any sequence of sequences, of int32
    // arrays, or a 2-D integer array — straight into the packed token buffer: the reference's signature copies a
    // std::vector<std::vector<int>> by value (fastsk.cpp:30), i.e. every token twice through pybind's generic casters before
    // the engine sees it; here a list row is one PyLong_AsLong per token and an int32 array row one memcpy.
    size_t add(const py::handle& X) {
        if (py::isinstance<py::array>(X)) {
            auto a = py::array_t<int32_t, py::array::c_style | py::array::forcecast>::ensure(X);
            if (a && a.ndim() == 2) {
                const py::ssize_t n = a.shape(0), L = a.shape(1);
                tokens.insert(tokens.end(), a.data(), a.data() + n * L);
                for (py::ssize_t i = 0; i < n; ++i) offsets.push_back(offsets.back() + L);
                return (size_t)n;
            }
        }
        PyObject* seq = PySequence_Fast(X.ptr(), "expected a sequence of token sequences");
        if (!seq) throw py::error_already_set();
        py::object hold = py::reinterpret_steal<py::object>(seq);
        const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
        PyObject** rows = PySequence_Fast_ITEMS(seq);
        {   // one reservation for every row whose length is cheap to ask for
            size_t total = tokens.size();
            for (Py_ssize_t i = 0; i < n; ++i) {
                const Py_ssize_t len = PyObject_Length(rows[i]);
                if (len < 0) { PyErr_Clear(); continue; }
                total += (size_t)len;
            }
            tokens.reserve(total);
        }
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject* r = rows[i];
            if (PyList_CheckExact(r) || PyTuple_CheckExact(r)) {
                const Py_ssize_t len = PySequence_Fast_GET_SIZE(r);
                PyObject** it = PySequence_Fast_ITEMS(r);
                const size_t at = tokens.size();
                tokens.resize(at + (size_t)len);
                int32_t* dst = tokens.data() + at;
                for (Py_ssize_t q = 0; q < len; ++q) {
                    const long v = PyLong_AsLong(it[q]);
                    if (v == -1 && PyErr_Occurred()) throw py::error_already_set();
                    if (v < INT32_MIN || v > INT32_MAX) throw py::value_error("token ids must fit 32 bits");
                    dst[q] = (int32_t)v;
                }
            } else {  // an array (or any other sequence) row
                auto a = py::array_t<int32_t, py::array::c_style | py::array::forcecast>::ensure(py::handle(r));
                if (!a || a.ndim() != 1) { PyErr_Clear(); throw py::value_error("every sequence must be a list, a tuple or a 1-D array of token ids"); }
                tokens.insert(tokens.end(), a.data(), a.data() + a.shape(0));
            }
            offsets.push_back((int64_t)tokens.size());
        }
        return (size_t)n;
    }
};

// ---- DLPack (the ABI of dlpack.h v0.8, restated: nothing else of it is needed here) --------------------
struct DLDevice { int32_t device_type; int32_t device_id; };
struct DLDataType { uint8_t code; uint8_t bits; uint16_t lanes; };
struct DLTensor {
    void* data; DLDevice device; int32_t ndim; DLDataType dtype; int64_t* shape; int64_t* strides; uint64_t byte_offset;
};
struct DLManagedTensor { DLTensor dl_tensor; void* manager_ctx; void (*deleter)(DLManagedTensor*); };
constexpr int32_t kDLROCM = 10;
constexpr uint8_t kDLFloat = 2;
struct BlockOwner { int64_t shape[2]; };   // manager_ctx of a block handed out through DLPack

int parse_collective(const std::string& c) {
    if (c == "auto") return FSK_COLL_AUTO;
    if (c == "rccl") return FSK_COLL_RCCL;
    if (c == "p2p") return FSK_COLL_P2P;
    throw std::invalid_argument("collective must be 'auto', 'rccl' or 'p2p'");
}

int parse_path(const std::string& p) {
    if (p == "auto") return FSK_PATH_AUTO;
    if (p == "dense") return FSK_PATH_DENSE;
    if (p == "sparse") return FSK_PATH_SPARSE;
    throw std::invalid_argument("path must be 'auto', 'dense' or 'sparse'");
}

class FastSK {
    fsk_engine* h_ = nullptr;
    int64_t n_train_ = 0, n_test_ = 0;
    bool computed_ = false;
    int device0_ = 0;
    // skip_test_block="lazy" / None: the test x test block is left out of compute_kernel and computed only if asked for
    bool lazy_test_block_ = false, test_block_missing_ = false;
    std::vector<int32_t> kept_tokens_;   // the call's input, kept while the test x test block is missing
    std::vector<int64_t> kept_offsets_;

    void check(int rc) const {
        if (rc == FSK_OK) return;
        std::string msg = fsk_last_error(h_);
        if (rc == FSK_EINVAL || rc == FSK_ESHORT) throw py::value_error(msg);
        throw std::runtime_error(msg);
    }
    void run_flat(const int32_t* tokens, const int64_t* offsets, int64_t n_train, int64_t n_test) {
        const bool skip = lazy_test_block_ && n_test > 0;
        if (lazy_test_block_) check(fsk_set_skip_test_block(h_, skip ? 1 : 0));
        int rc;
        {
            py::gil_scoped_release nogil;  // the reference holds the GIL for the whole call
            rc = fsk_compute(h_, tokens, offsets, n_train, n_test);
        }
        check(rc);
        n_train_ = n_train;
        n_test_ = n_test;
        computed_ = true;
        test_block_missing_ = skip;
        if (skip) {
            if (tokens != kept_tokens_.data()) {
                kept_tokens_.assign(tokens + offsets[0], tokens + offsets[n_train + n_test]);
                kept_offsets_.assign(offsets, offsets + n_train + n_test + 1);
                for (auto& o : kept_offsets_) o -= offsets[0];
            }
        } else {
            kept_tokens_.clear(); kept_tokens_.shrink_to_fit();
            kept_offsets_.clear();
        }
    }
    void run(const Flat& f, int64_t n_train, int64_t n_test) { run_flat(f.tokens.data(), f.offsets.data(), n_train, n_test); }
    // something wants cells with both sequences in the test set: the same call again, nothing left out
    void need_test_block() {
        if (!test_block_missing_) return;
        check(fsk_set_skip_test_block(h_, 0));
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = fsk_compute(h_, kept_tokens_.data(), kept_offsets_.data(), n_train_, n_test_);
        }
        check(rc);
        test_block_missing_ = false;
        kept_tokens_.clear(); kept_tokens_.shrink_to_fit();
        kept_offsets_.clear();
    }
    // device memory of the engine's first GPU (released with fsk_free_device) as a DLPack capsule of float64
    py::capsule dlpack_of(double* dev, int ndim, int64_t d0, int64_t d1) {
        auto* owner = new BlockOwner{{d0, d1}};
        auto* mt = new DLManagedTensor{};
        mt->dl_tensor.data = dev;
        mt->dl_tensor.device = DLDevice{kDLROCM, device0_};
        mt->dl_tensor.ndim = ndim;
        mt->dl_tensor.dtype = DLDataType{kDLFloat, 64, 1};
        mt->dl_tensor.shape = owner->shape;
        mt->dl_tensor.strides = nullptr;  // compact row-major
        mt->dl_tensor.byte_offset = 0;
        mt->manager_ctx = owner;
        mt->deleter = [](DLManagedTensor* t) {  // (the block outlives the FastSK object if its consumer does)
            (void)fsk_free_device(nullptr, t->dl_tensor.data);
            delete static_cast<BlockOwner*>(t->manager_ctx);
            delete t;
        };
        return py::capsule(mt, "dltensor", [](PyObject* cap) {
            if (PyCapsule_IsValid(cap, "dltensor")) {  // never consumed: ours to free
                auto* t = static_cast<DLManagedTensor*>(PyCapsule_GetPointer(cap, "dltensor"));
                if (t && t->deleter) t->deleter(t);
            }
        });
    }
    py::capsule dlpack_block(int64_t i0, int64_t i1, int64_t j0, int64_t j1) {
        if (!computed_) throw std::runtime_error("call compute_kernel or compute_train first");
        if (i1 < i0 || j1 < j0) throw py::value_error("empty block");
        if (i1 > n_train_ && j1 > n_train_) need_test_block();
        double* dev = nullptr;
        check(fsk_alloc_block_device(h_, i0, i1, j0, j1, &dev));
        return dlpack_of(dev, 2, i1 - i0, j1 - j0);
    }
    py::array_t<double> block(bool test) const {
        if (!computed_) throw std::runtime_error("call compute_kernel or compute_train first");
        const int64_t rows = test ? n_test_ : n_train_;
        py::array_t<double> out({(py::ssize_t)rows, (py::ssize_t)n_train_});
        if (rows > 0) check(test ? fsk_get_test(h_, out.mutable_data()) : fsk_get_train(h_, out.mutable_data()));
        return out;
    }
    // the reference's return type, vector<vector<double>> -> a list of lists of floats, built from the block's one host copy
    // (no vector of vectors in between; what is left is one PyFloat per cell: ~12 ns each, 8 * 10^6 of them at EP300)
    static py::list to_lists(const py::array_t<double>& a) {
        const py::ssize_t r = a.shape(0), c = a.shape(1);
        const double* p = a.data();
        py::list out(r);
        for (py::ssize_t i = 0; i < r; ++i) {
            PyObject* row = PyList_New(c);
            if (!row) throw py::error_already_set();
            for (py::ssize_t j = 0; j < c; ++j) {
                PyObject* v = PyFloat_FromDouble(p[i * c + j]);
                if (!v) { Py_DECREF(row); throw py::error_already_set(); }
                PyList_SET_ITEM(row, j, v);
            }
            PyList_SET_ITEM(out.ptr(), i, row);  // (py::list(r) holds r null slots)
        }
        return out;
    }

public:
    FastSK(int g, int m, int t, bool approx, double delta, int max_iters, bool skip_variance, int device,
           const std::string& path, py::object seed, py::object skip_test_block, py::object devices,
           const std::string& collective, int deadline_ms) {
        fsk_config c{};
        if (skip_test_block.is_none()) lazy_test_block_ = true;
        else if (py::isinstance<py::str>(skip_test_block)) {
            if (skip_test_block.cast<std::string>() != "lazy") throw py::value_error("skip_test_block must be False, True or \"lazy\"");
            lazy_test_block_ = true;
        }
        c.skip_test_block = lazy_test_block_ ? 0 : (skip_test_block.cast<bool>() ? 1 : 0);
        c.g = g; c.m = m; c.t = t; c.approx = approx; c.delta = delta; c.max_iters = max_iters;
        c.skip_variance = skip_variance; c.device = device; c.path = parse_path(path);
        c.collective = parse_collective(collective);
        c.deadline_ms = deadline_ms;
        int rc;
        if (devices.is_none()) {
            device0_ = device;
            rc = fsk_create(&c, &h_);
        } else {
            const std::vector<int32_t> devs = devices.cast<std::vector<int32_t>>();
            if (devs.empty()) throw py::value_error("devices must list at least one GPU");
            device0_ = devs[0];
            rc = fsk_create_multi(&c, devs.data(), (int32_t)devs.size(), &h_);
        }
        if (rc != FSK_OK) {
            std::string msg = fsk_last_error(nullptr);
            if (rc == FSK_EINVAL) throw py::value_error(msg);
            throw std::runtime_error(msg);
        }
        if (!seed.is_none()) check(fsk_set_seed(h_, seed.cast<uint64_t>()));
    }
    ~FastSK() { fsk_destroy(h_); }
    FastSK(const FastSK&) = delete;
    FastSK& operator=(const FastSK&) = delete;

    // FastSK::compute_kernel, fastsk.cpp:30-118. Xtrain / Xtest: lists of lists of ints as the reference takes them — or
    // tuples, int arrays per sequence, one 2-D integer array (fixed-length sequences) — see Flat::add
    void compute_kernel(py::object Xtrain, py::object Xtest) {
        Flat f;
        const size_t ntr = f.add(Xtrain), nte = f.add(Xtest);
        if (ntr == 0 || nte == 0) throw py::value_error("Xtrain and Xtest must be non-empty (use compute_train for train only)");
        run(f, (int64_t)ntr, (int64_t)nte);
    }
    // Additive: ragged sequences already flattened (tokens + offsets, train rows first), e.g.
    // from FastaUtility.read_packed
    void compute_kernel_flat(py::array_t<int32_t, py::array::c_style> tokens, py::array_t<int64_t, py::array::c_style> offsets,
                             int64_t n_train) {
        if (tokens.ndim() != 1 || offsets.ndim() != 1 || offsets.shape(0) < 2) throw py::value_error("expected 1-D tokens and offsets");
        const int64_t n = (int64_t)offsets.shape(0) - 1;
        if (n_train <= 0 || n_train > n) throw py::value_error("n_train out of range");
        if (offsets.data()[0] != 0 || offsets.data()[n] != (int64_t)tokens.shape(0)) throw py::value_error("offsets must start at 0 and end at len(tokens)");
        run_flat(tokens.data(), offsets.data(), n_train, n - n_train);
    }
    // FastSK::compute_train, fastsk.cpp:120-188
    void compute_train(py::object Xtrain) {
        Flat f;
        const size_t ntr = f.add(Xtrain);
        if (ntr == 0) throw py::value_error("Xtrain must be non-empty");
        run(f, (int64_t)ntr, 0);
    }
    py::list get_train_kernel() const { return to_lists(block(false)); }  // fastsk.cpp:190-200
    py::list get_test_kernel() const { return to_lists(block(true)); }    // fastsk.cpp:202-217
    py::array_t<double> get_train_kernel_np() const { return block(false); }
    py::array_t<double> get_test_kernel_np() const { return block(true); }
    std::vector<double> get_stdevs() const {  // fastsk.cpp:219-221
        int32_t n = 0;
        check(fsk_get_stdevs(h_, nullptr, 0, &n));
        std::vector<double> v((size_t)n);
        if (n) check(fsk_get_stdevs(h_, v.data(), n, &n));
        return v;
    }
    void save_kernel(const std::string& path) {  // fastsk.cpp:223-237 (the whole N x N matrix)
        need_test_block();
        check(fsk_save_kernel(h_, path.c_str()));
    }
    // the same blocks as device-resident DLPack capsules (float64, row-major, on the engine's first GPU):
    // torch.from_dlpack(f.get_train_kernel_dlpack()) keeps the kernel matrix on the GPU for the SVM stage
    py::capsule get_train_kernel_dlpack() { return dlpack_block(0, n_train_, 0, n_train_); }
    py::capsule get_test_kernel_dlpack() { return dlpack_block(n_train_, n_train_ + n_test_, 0, n_train_); }
    py::capsule get_block_dlpack(int64_t i0, int64_t i1, int64_t j0, int64_t j1) { return dlpack_block(i0, i1, j0, j1); }
    void set_combo_order(std::vector<int32_t> order) {
        // a result whose test x test block is still to come belongs to the OLD order: complete it first, so that
        // every block of one compute_kernel call comes from one set of combos
        need_test_block();
        check(fsk_set_combo_order(h_, order.data(), (int32_t)order.size()));
    }
    py::array_t<double> get_block(int64_t i0, int64_t i1, int64_t j0, int64_t j1) {
        if (i1 < i0 || j1 < j0) throw py::value_error("empty block");
        if (i1 > n_train_ && j1 > n_train_) need_test_block();
        py::array_t<double> out({(py::ssize_t)(i1 - i0), (py::ssize_t)(j1 - j0)});
        check(fsk_get_block(h_, i0, i1, j0, j1, out.mutable_data()));
        return out;
    }
    py::array_t<uint64_t> get_counts_np() {
        if (!computed_) throw std::runtime_error("call compute_kernel or compute_train first");
        need_test_block();
        const int64_t N = n_train_ + n_test_;
        py::array_t<uint64_t> out((py::ssize_t)(N * (N + 1) / 2));
        check(fsk_get_counts(h_, out.mutable_data()));
        return out;
    }
    // ---- additive: the raw integer kernel without copying the triangle out (exact and skip-variance modes).
    // The reference has no such getter (its K is normalised in place, fastsk_kernel.cpp:96-103); these are what a
    // caller uses to check a 100k-sequence result, whose triangle is 40 GB.
    // (sum, xor of cell * (index | 1)) mod 2^64 over the integer cells of rows [row_begin, row_end): fsk_counts_digest
    py::tuple counts_digest(int64_t row_begin, int64_t row_end) {
        if (!computed_) throw std::runtime_error("call compute_kernel or compute_train first");
        const int64_t N = n_train_ + n_test_;
        if (row_end < 0) row_end = N;
        if (row_end > n_train_ + 1) need_test_block();  // (row n_train holds one test x test cell, its diagonal, always computed)
        uint64_t d[2] = {0, 0};
        check(fsk_counts_digest(h_, row_begin, row_end, d));
        return py::make_tuple(py::int_(d[0]), py::int_(d[1]));
    }
    py::array_t<uint64_t> get_counts_block(int64_t i0, int64_t i1, int64_t j0, int64_t j1) {
        if (!computed_) throw std::runtime_error("call compute_kernel or compute_train first");
        if (i1 < i0 || j1 < j0) throw py::value_error("empty block");
        if (i1 > n_train_ && j1 > n_train_) need_test_block();
        py::array_t<uint64_t> out({(py::ssize_t)(i1 - i0), (py::ssize_t)(j1 - j0)});
        check(fsk_get_counts_block(h_, i0, i1, j0, j1, out.mutable_data()));
        return out;
    }
    // scattered cells (rows[q], cols[q]) of the symmetric integer matrix: tri_access of arbitrary pairs, shared.cpp:97-117
    py::array_t<uint64_t> get_counts_cells(py::array_t<int64_t, py::array::c_style | py::array::forcecast> rows,
                                           py::array_t<int64_t, py::array::c_style | py::array::forcecast> cols) {
        if (!computed_) throw std::runtime_error("call compute_kernel or compute_train first");
        if (rows.ndim() != 1 || cols.ndim() != 1 || rows.shape(0) != cols.shape(0)) throw py::value_error("rows and cols must be 1-D and of equal length");
        const int64_t n = (int64_t)rows.shape(0);
        if (test_block_missing_)
            for (int64_t q = 0; q < n; ++q)
                if (rows.data()[q] >= n_train_ && cols.data()[q] >= n_train_ && rows.data()[q] != cols.data()[q]) { need_test_block(); break; }
        py::array_t<uint64_t> out((py::ssize_t)n);
        check(fsk_get_counts_cells(h_, rows.data(), cols.data(), n, out.mutable_data()));
        return out;
    }
    // the reference's K itself — the whole normalised lower triangle, cell (i, j <= i) at i(i+1)/2 + j
    // (fastsk_kernel.cpp:96-103) — resident on the GPU as a 1-D float64 DLPack capsule
    py::capsule get_triangle_dlpack() {
        if (!computed_) throw std::runtime_error("call compute_kernel or compute_train first");
        need_test_block();
        double* dev = nullptr;
        check(fsk_alloc_triangle_device(h_, &dev));
        const int64_t N = n_train_ + n_test_;
        return dlpack_of(dev, 1, N * (N + 1) / 2, 1);
    }
    py::dict stats() const {
        fsk_stats s;
        check(fsk_get_stats(h_, &s));
        py::dict d;
        d["n_seq"] = s.n_seq; d["n_feat"] = s.n_feat; d["n_pairs"] = s.n_pairs; d["alphabet"] = s.alphabet;
        d["bits_per_symbol"] = s.bits_per_symbol; d["key_space"] = s.key_space;
        d["path_used"] = s.path_used == FSK_PATH_DENSE ? "dense" : "sparse";
        d["n_combos_total"] = s.n_combos_total; d["combos_done"] = s.combos_done;
        d["cell_updates"] = s.cell_updates; d["launches"] = s.launches;
        d["test_block_computed"] = computed_ && !test_block_missing_;
        fsk_multi_info mi;
        check(fsk_get_multi_info(h_, &mi));
        py::list devs;
        for (int r = 0; r < mi.ndev; ++r) devs.append(mi.devices[r]);
        if (mi.ndev == 0) devs.append(device0_);
        d["devices"] = devs;
        d["collective"] = mi.ndev == 0 ? "none" : mi.collective == FSK_COLL_RCCL ? "rccl" : "p2p";
        d["comm_ranks"] = mi.comm_ranks; d["exchange_bands"] = mi.bands; d["exchange_int32"] = (bool)mi.narrow;
        return d;
    }
    void fit(double, double, double, const std::string&) const {
        PyErr_SetString(PyExc_NotImplementedError,
                        "fit(): the LIBSVM stage is outside the accelerated path; feed get_train_kernel() to scikit-learn "
                        "as the reference's own pipeline does (test/run_check.py:48-61)");
        throw py::error_already_set();
    }
    double score(const std::string&) const {
        PyErr_SetString(PyExc_NotImplementedError, "score(): see fit()");
        throw py::error_already_set();
    }
};

}  // namespace

PYBIND11_MODULE(_fastsk, m) {
    m.doc() = "MI355X-native gapped-k-mer kernel engine behind the FastSK Python surface";
    py::class_<FastSK>(m, "FastSK")
        .def(py::init<int, int, int, bool, double, int, bool, int, const std::string&, py::object, py::object, py::object,
                      const std::string&, int>(),
             py::arg("g"), py::arg("m"), py::arg("t") = -1, py::arg("approx") = false, py::arg("delta") = 0.025,
             py::arg("max_iters") = -1, py::arg("skip_variance") = false, py::arg("device") = 0,
             py::arg("path") = "auto", py::arg("seed") = py::none(), py::arg("skip_test_block") = false,
             py::arg("devices") = py::none(), py::arg("collective") = "auto", py::arg("deadline_ms") = 0)
        .def("compute_kernel", &FastSK::compute_kernel, py::arg("Xtrain"), py::arg("Xtest"))
        .def("compute_kernel_flat", &FastSK::compute_kernel_flat, py::arg("tokens").noconvert(), py::arg("offsets").noconvert(),
             py::arg("n_train"))
        .def("compute_train", &FastSK::compute_train, py::arg("Xtrain"))
        .def("get_train_kernel", &FastSK::get_train_kernel)
        .def("get_test_kernel", &FastSK::get_test_kernel)
        .def("get_stdevs", &FastSK::get_stdevs)
        .def("save_kernel", &FastSK::save_kernel)
        .def("fit", &FastSK::fit, py::arg("C") = 1.0, py::arg("nu") = 0.5, py::arg("eps") = 0.001,
             py::arg("kernel_type") = "linear")
        .def("score", &FastSK::score, py::arg("metric") = "auc")
        // additive, non-breaking extras
        .def("get_train_kernel_np", &FastSK::get_train_kernel_np)
        .def("get_test_kernel_np", &FastSK::get_test_kernel_np)
        .def("get_block", &FastSK::get_block, py::arg("i0"), py::arg("i1"), py::arg("j0"), py::arg("j1"))
        .def("get_train_kernel_dlpack", &FastSK::get_train_kernel_dlpack)
        .def("get_test_kernel_dlpack", &FastSK::get_test_kernel_dlpack)
        .def("get_block_dlpack", &FastSK::get_block_dlpack, py::arg("i0"), py::arg("i1"), py::arg("j0"), py::arg("j1"))
        .def("get_counts_np", &FastSK::get_counts_np)
        .def("counts_digest", &FastSK::counts_digest, py::arg("row_begin") = 0, py::arg("row_end") = -1)
        .def("get_counts_block", &FastSK::get_counts_block, py::arg("i0"), py::arg("i1"), py::arg("j0"), py::arg("j1"))
        .def("get_counts_cells", &FastSK::get_counts_cells, py::arg("rows"), py::arg("cols"))
        .def("get_triangle_dlpack", &FastSK::get_triangle_dlpack)
        .def("set_combo_order", &FastSK::set_combo_order, py::arg("order"))
        .def("stats", &FastSK::stats);
    m.attr("__version__") = "dev";
    m.attr("abi_version") = fsk_abi_version();
}
