// ilupp_amd/csrc/pybind_module.cpp -- the pybind11 shim over the C ABI (include/ilupp_hip.h): a compiled module with the
// surface of the reference's `ilupp._ilupp` for the hot path (src/binding.cpp:233-264 object members, :279 index_size,
// :299-310 ILUTPreconditioner, :313-326 ILUTPPreconditioner, :343-356 ILUCPPreconditioner with :178-196 permutations(),
// :366-397 the ILU0 / IChol0 / ICholT factories, :399-447 the stand-alone factor functions).
// Host code only: buffer checks, GIL handling, numpy egress; all arithmetic happens behind the C ABI on the GPU.
//   g++ -O2 -shared -fPIC $(python3 -m pybind11 --includes) pybind_module.cpp -L.. -lilupp_hip -Wl,-rpath,'$ORIGIN'
//       -o ../_ilupp_hip$(python3-config --extension-suffix)
#include <pybind11/pybind11.h>
#include <pybind11/numpy.h>

#include <string.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ilupp_hip.h"

namespace py = pybind11;

namespace {

[[noreturn]] void fail(int) { throw std::runtime_error(ilupp_hip_last_error()); }       // pybind11: std::exception -> RuntimeError
inline void ok(int rc) { if (rc) fail(rc); }

// what make_matrix / make_vector demand of a buffer (binding.cpp:33-98), with the reference's messages
py::buffer_info flat(const py::buffer &b, const char *name)
{
    py::buffer_info info = b.request();
    if (info.ndim != 1) throw std::runtime_error(std::string("Expected 1D array for ") + name + "!");
    if (info.strides[0] != info.itemsize) throw std::runtime_error(std::string("Expected contiguous array for ") + name + "!");
    return info;
}
py::buffer_info reals(const py::buffer &b, const char *name)
{
    py::buffer_info info = flat(b, name);
    if (info.format != py::format_descriptor<double>::format())
        throw std::runtime_error(std::string("Expected d (d) array for ") + name + ", got " + info.format + "!");
    return info;
}
py::buffer_info ints(const py::buffer &b, const char *name)
{
    py::buffer_info info = flat(b, name);
    const bool integral = info.format.size() == 1 && std::string("ilq").find(info.format[0]) != std::string::npos;
    if (!integral || info.itemsize != (py::ssize_t)sizeof(int32_t))
        throw std::runtime_error(std::string("Expected integer type with length 4 for ") + name + ", got " + info.format + "!");
    return info;
}

struct Csr { const double *val; const int32_t *idx, *ptr; int32_t n; int row_major; };
Csr borrow(const py::buffer &data, const py::buffer &indices, const py::buffer &indptr, bool is_csr)
{
    const py::buffer_info d = reals(data, "A_data"), i = ints(indices, "A_indices"), p = ints(indptr, "A_indptr");
    if (p.shape[0] <= 1) throw std::runtime_error("matrix has size 0!");
    if (i.shape[0] != d.shape[0]) throw std::runtime_error("indices and data should have the same size!");
    return Csr{static_cast<const double *>(d.ptr), static_cast<const int32_t *>(i.ptr), static_cast<const int32_t *>(p.ptr),
               (int32_t)(p.shape[0] - 1), is_csr ? 1 : 0};
}

// a factorisation in HBM; LU and LLT are distinct Python classes like the reference's Generic*Preconditioner
struct Factorisation {
    ilupp_precond *h = nullptr;
    Factorisation() = default;
    Factorisation(const Factorisation &) = delete;
    Factorisation(Factorisation &&o) noexcept : h(o.h) { o.h = nullptr; }
    ~Factorisation() { if (h) ilupp_hip_destroy(h); }
};
struct LU : Factorisation { using Factorisation::Factorisation; };
struct LLT : Factorisation { using Factorisation::Factorisation; };
struct ILUT : Factorisation { using Factorisation::Factorisation; };
struct ILUC : Factorisation { using Factorisation::Factorisation; };

template <class Make>
ilupp_precond *build(Make make)
{
    ilupp_precond *h = nullptr;
    int rc;
    {
        py::gil_scoped_release release;              // factorisation runs without the GIL (binding.cpp:371)
        rc = make(&h);
    }
    ok(rc);
    return h;
}

py::list egress(const Factorisation &f)
{
    py::list out;
    const int nf = ilupp_hip_num_factors(f.h);
    for (int k = 0; k < nf; ++k) {
        int32_t rows = 0, cols = 0; int64_t nnz = 0; int row_major = 1;
        ok(ilupp_hip_factor_info(f.h, k, &rows, &cols, &nnz, &row_major));
        py::array_t<double> data(nnz);
        py::array_t<int32_t> indices(nnz), indptr(rows + 1);
        ok(ilupp_hip_factor_copy(f.h, k, data.mutable_data(), indices.mutable_data(), indptr.mutable_data()));
        out.append(py::make_tuple(data, indices, indptr, row_major != 0, rows, cols));
    }
    return out;
}

void solve_in_place(const Factorisation &f, const py::buffer &x, bool transposed)
{
    py::buffer_info v = reals(x, "b");
    if (v.readonly) throw std::runtime_error("b must be writable");
    if (v.shape[0] != ilupp_hip_dimension(f.h)) throw std::runtime_error("vector has wrong size for preconditioner!");
    ok(transposed ? ilupp_hip_apply_trans(f.h, static_cast<double *>(v.ptr), v.shape[0])
                  : ilupp_hip_apply(f.h, static_cast<double *>(v.ptr), v.shape[0]));       // GIL held, as binding.cpp:237-254
}

template <class T>
py::class_<T> members(py::module_ &m, const char *name)
{
    return py::class_<T>(m, name)
        .def("apply", [](const T &f, py::buffer x) { solve_in_place(f, x, false); })
        .def("apply_trans", [](const T &f, py::buffer x) { solve_in_place(f, x, true); })
        .def_property_readonly("total_nnz", [](const T &f) { return ilupp_hip_total_nnz(f.h); })
        .def("factors_info", [](const T &f) { return egress(f); })
        .def_property_readonly("memory_used_calculations", [](const T &f) { return ilupp_hip_memory_used_calculations(f.h); })
        .def_property_readonly("memory_allocated_calculations", [](const T &f) { return ilupp_hip_memory_allocated_calculations(f.h); })
        .def_property_readonly("memory", [](const T &f) { return ilupp_hip_memory(f.h); })
        .def_property_readonly("exists", [](const T &f) { return ilupp_hip_exists(f.h) != 0; })
        .def_property_readonly("special_info", [](const T &f) { return std::string(ilupp_hip_special_info(f.h)); })
        .def("print_info", [](const T &f) { ilupp_hip_print_info(f.h); });
}

template <class T> T adopt(ilupp_precond *h) { T t; t.h = h; return t; }

// the multilevel preconditioner (binding.cpp:284-298).  `param`: an object of ilupp_amd.params.iluplusplus_precond_parameter -- the
// reference's parameter class restated in Python -- whose _to_ml_params() decides whether the engine has the family and fills the
// C-ABI block (ctypes structure with the layout of ilupp_ml_params); inside the reference this would be iluplusplus_precond_parameter
// itself and the function `to_block` of INTEGRATION.md
struct Multilevel {
    ilupp_ml *h = nullptr;
    int32_t n = 0;
    Multilevel() = default;
    Multilevel(const Multilevel &) = delete;
    Multilevel(Multilevel &&o) noexcept : h(o.h), n(o.n) { o.h = nullptr; }
    ~Multilevel() { if (h) ilupp_hip_ml_destroy(h); }
};

ilupp_ml_params block_of(const py::object &param)
{
    const py::object blk = param.attr("_to_ml_params")();                 // raises NotImplementedError outside the built family
    const py::bytes raw = py::module_::import("builtins").attr("bytes")(blk);
    const std::string bytes = raw;
    if (bytes.size() != sizeof(ilupp_ml_params)) throw std::runtime_error("parameter block of unexpected size");
    ilupp_ml_params p;
    memcpy(&p, bytes.data(), sizeof(p));
    return p;
}

void ml_solve_in_place(const Multilevel &f, const py::buffer &x, int transposed)
{
    py::buffer_info v = reals(x, "b");
    if (v.readonly) throw std::runtime_error("b must be writable");
    if (v.shape[0] != f.n) throw std::runtime_error("vector has wrong size for preconditioner!");
    ok(ilupp_hip_ml_apply(f.h, static_cast<double *>(v.ptr), v.shape[0], transposed));
}

// the two factorisations with column pivoting (binding.cpp:313-326 ILUTPPreconditioner, :343-356 ILUCPPreconditioner): one kind of object
// behind the C ABI; `by_rows`: ILUTP2's factors belong to the rows of the view, ILUCP4's L to its columns
struct Pivoted {
    ilupp_ilucp *h = nullptr;
    int32_t n = 0;
    bool csr = true, by_rows = false;
    Pivoted() = default;
    Pivoted(const Pivoted &) = delete;
    Pivoted(Pivoted &&o) noexcept : h(o.h), n(o.n), csr(o.csr), by_rows(o.by_rows) { o.h = nullptr; }
    ~Pivoted() { if (h) ilupp_hip_ilucp_destroy(h); }
};
struct ILUTP : Pivoted { using Pivoted::Pivoted; };
struct ILUCP : Pivoted { using Pivoted::Pivoted; };

struct PivotedArrays { py::array_t<double> ld, ud; py::array_t<int32_t> li, lp, ui, up, perm; };
PivotedArrays pivoted_arrays(const Pivoted &f)
{
    int32_t n = 0; int64_t nl = 0, nu = 0;
    ok(ilupp_hip_ilucp_info(f.h, &n, &nl, &nu, nullptr));
    PivotedArrays a{py::array_t<double>(nl), py::array_t<double>(nu), py::array_t<int32_t>(nl), py::array_t<int32_t>(n + 1),
                    py::array_t<int32_t>(nu), py::array_t<int32_t>(n + 1), py::array_t<int32_t>(n)};
    ok(ilupp_hip_ilucp_copy(f.h, a.ld.mutable_data(), a.li.mutable_data(), a.lp.mutable_data(), a.ud.mutable_data(), a.ui.mutable_data(),
                            a.up.mutable_data(), a.perm.mutable_data()));
    return a;
}

// [left, right] as the classes hold them (preconditioner_implementation.h:1050-1078, :1117-1147): the factors of the major-order view change
// sides and labels when the class factorised the transposed matrix (transpose_in_place)
py::list pivoted_factors(const Pivoted &f)
{
    const PivotedArrays a = pivoted_arrays(f);
    const py::tuple L_rows = py::make_tuple(a.ld, a.li, a.lp, true, f.n, f.n), L_cols = py::make_tuple(a.ld, a.li, a.lp, false, f.n, f.n);
    const py::tuple U_rows = py::make_tuple(a.ud, a.ui, a.up, true, f.n, f.n), U_cols = py::make_tuple(a.ud, a.ui, a.up, false, f.n, f.n);
    py::list out;
    if (f.by_rows) {
        if (f.csr) { out.append(L_rows); out.append(U_rows); } else { out.append(U_cols); out.append(L_cols); }
    } else {
        if (!f.csr) { out.append(L_cols); out.append(U_rows); } else { out.append(U_cols); out.append(L_rows); }
    }
    return out;
}

// binding.cpp:178-196: (left, right) -- the permutation stands on the side of the factor U came from
py::tuple pivoted_permutations(const Pivoted &f)
{
    const py::object perm = pivoted_arrays(f).perm;
    const bool right = f.by_rows ? f.csr : !f.csr;
    if (right) return py::make_tuple(py::none(), perm);
    return py::make_tuple(perm, py::none());
}

template <class T, class Create>
T make_pivoted(Create create, bool by_rows, py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t max_fill_in,
               double threshold, double piv_tol, int32_t row_pos, double mem_factor)
{
    const Csr a = borrow(data, indices, indptr, is_csr);
    T f;
    f.n = a.n; f.csr = is_csr; f.by_rows = by_rows;
    int rc;
    {
        py::gil_scoped_release release;              // binding.cpp:319-321, :349-351
        rc = create(a.val, a.idx, a.ptr, a.n, a.row_major, max_fill_in, threshold, piv_tol, row_pos, mem_factor, &f.h);
    }
    ok(rc);
    return f;
}

template <class T>
py::class_<T> pivoted_members(py::module_ &m, const char *name)
{
    return py::class_<T>(m, name)
        .def("apply", [](const T &f, py::buffer x) {
            py::buffer_info v = reals(x, "b");
            if (v.readonly) throw std::runtime_error("b must be writable");
            if (v.shape[0] != f.n) throw std::runtime_error("vector has wrong size for preconditioner!");
            ok(ilupp_hip_ilucp_apply(f.h, static_cast<double *>(v.ptr), v.shape[0], 0));
        })
        .def("apply_trans", [](const T &f, py::buffer x) {
            py::buffer_info v = reals(x, "b");
            if (v.readonly) throw std::runtime_error("b must be writable");
            if (v.shape[0] != f.n) throw std::runtime_error("vector has wrong size for preconditioner!");
            ok(ilupp_hip_ilucp_apply(f.h, static_cast<double *>(v.ptr), v.shape[0], 1));
        })
        .def_property_readonly("total_nnz", [](const T &f) { return ilupp_hip_ilucp_total_nnz(f.h); })
        .def_property_readonly("zero_pivots", [](const T &f) { return ilupp_hip_ilucp_zero_pivots(f.h); })
        .def("factors_info", [](const T &f) { return pivoted_factors(f); })
        .def("permutations", [](const T &f) { return pivoted_permutations(f); })
        .def_property_readonly("memory_used_calculations", [](const T &) { return 0.0; })
        .def_property_readonly("memory_allocated_calculations", [](const T &) { return 0.0; })
        .def_property_readonly("memory", [](const T &) { return 0.0; })
        .def_property_readonly("exists", [](const T &) { return true; })
        .def_property_readonly("special_info", [](const T &) { return std::string(); })
        .def("print_info", [](const T &f) {
            py::print("An incomplete LU factorisation with column pivoting:", ilupp_hip_ilucp_total_nnz(f.h), "entries");
        });
}

}  // namespace

PYBIND11_MODULE(_ilupp_hip, m)
{
    m.doc() = "MI355X backend of the ilupp preconditioners (pybind11 shim over the C ABI of include/ilupp_hip.h)";
    m.def("index_size", []() { return ilupp_hip_index_size(); });
    m.def("device_count", []() { return ilupp_hip_device_count(); });

    members<LU>(m, "GenericLUPreconditioner");
    members<LLT>(m, "GenericLLTPreconditioner");
    members<ILUT>(m, "ILUTPreconditioner")
        .def(py::init([](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t max_fill_in, double threshold) {
            const Csr a = borrow(data, indices, indptr, is_csr);
            return adopt<ILUT>(build([&](ilupp_precond **h) { return ilupp_hip_ilut_create(a.val, a.idx, a.ptr, a.n, a.row_major, max_fill_in, threshold, h); }));
        }));

    members<ILUC>(m, "ILUCPreconditioner")
        .def(py::init([](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t max_fill_in, double threshold) {
            const Csr a = borrow(data, indices, indptr, is_csr);
            return adopt<ILUC>(build([&](ilupp_precond **h) { return ilupp_hip_iluc_create(a.val, a.idx, a.ptr, a.n, a.row_major, max_fill_in, threshold, h); }));
        }));

    pivoted_members<ILUTP>(m, "ILUTPPreconditioner")
        .def(py::init([](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t max_fill_in, double threshold,
                         double piv_tol, int32_t row_pos, double mem_factor) {
            return make_pivoted<ILUTP>(ilupp_hip_ilutp_create, true, data, indices, indptr, is_csr, max_fill_in, threshold, piv_tol, row_pos, mem_factor);
        }));
    pivoted_members<ILUCP>(m, "ILUCPPreconditioner")
        .def(py::init([](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t max_fill_in, double threshold,
                         double piv_tol, int32_t row_pos, double mem_factor) {
            return make_pivoted<ILUCP>(ilupp_hip_ilucp_create, false, data, indices, indptr, is_csr, max_fill_in, threshold, piv_tol, row_pos, mem_factor);
        }));

    py::class_<Multilevel>(m, "MultilevelILUCDPPreconditioner")
        .def(py::init([](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, py::object param) {
            const Csr a = borrow(data, indices, indptr, is_csr);
            const ilupp_ml_params p = block_of(param);
            Multilevel f;
            f.n = a.n;
            int rc;
            {
                py::gil_scoped_release release;          // binding.cpp:292-294
                rc = ilupp_hip_ml_create(a.val, a.idx, a.ptr, a.n, a.row_major, &p, &f.h);
            }
            if (rc == ILUPP_ERR_UNSUPPORTED) { PyErr_SetString(PyExc_NotImplementedError, ilupp_hip_last_error()); throw py::error_already_set(); }
            ok(rc);
            return f;
        }))
        .def("apply", [](const Multilevel &f, py::buffer x) { ml_solve_in_place(f, x, 0); })
        .def("apply_trans", [](const Multilevel &f, py::buffer x) { ml_solve_in_place(f, x, 1); })
        .def_property_readonly("total_nnz", [](const Multilevel &f) { return ilupp_hip_ml_total_nnz(f.h); })
        .def("levels", [](const Multilevel &f) { return ilupp_hip_ml_levels(f.h); })
        .def("factors_info", [](const Multilevel &) { return py::list(); })                        // "not implemented", binding.cpp:158-163
        .def_property_readonly("memory_used_calculations", [](const Multilevel &) { return 0.0; })
        .def_property_readonly("memory_allocated_calculations", [](const Multilevel &) { return 0.0; })
        .def_property_readonly("memory", [](const Multilevel &) { return 0.0; })
        .def_property_readonly("exists", [](const Multilevel &) { return true; })
        .def_property_readonly("special_info", [](const Multilevel &) { return std::string(); })
        .def("print_info", [](const Multilevel &f) {
            py::print("A multilevel incomplete LU factorisation:", ilupp_hip_ml_levels(f.h), "levels,", ilupp_hip_ml_total_nnz(f.h), "entries");
        });

    // many matrices side by side (include/ilupp_hip.h: ilupp_hip_ml_create_batch): the objects the constructor gives one at a time.  On an
    // error every object already built is destroyed; the message names the first failing matrix and its status.
    m.def("MultilevelILUCDPPreconditioner_batch", [](py::list matrices, bool is_csr, py::object param) {
        const ilupp_ml_params p = block_of(param);
        const int32_t cnt = (int32_t)matrices.size();
        std::vector<Csr> as;
        std::vector<const double *> D; std::vector<const int32_t *> I, P; std::vector<int32_t> N;
        for (py::handle h : matrices) {
            py::tuple t = py::reinterpret_borrow<py::tuple>(h);
            as.push_back(borrow(py::reinterpret_borrow<py::buffer>(t[0]), py::reinterpret_borrow<py::buffer>(t[1]), py::reinterpret_borrow<py::buffer>(t[2]), is_csr));
        }
        for (const Csr &a : as) { D.push_back(a.val); I.push_back(a.idx); P.push_back(a.ptr); N.push_back(a.n); }
        std::vector<ilupp_ml *> out((size_t)cnt, nullptr);
        std::vector<int32_t> status((size_t)cnt, 0);
        int rc = ILUPP_OK;
        if (cnt > 0) {
            py::gil_scoped_release release;
            rc = ilupp_hip_ml_create_batch(cnt, D.data(), I.data(), P.data(), N.data(), is_csr ? 1 : 0, &p, out.data(), status.data());
        }
        if (rc != ILUPP_OK) {
            const std::string msg = ilupp_hip_last_error();
            int first = -1;
            for (int32_t k = 0; k < cnt; ++k) { if (status[(size_t)k] != 0 && first < 0) first = k; if (out[(size_t)k]) ilupp_hip_ml_destroy(out[(size_t)k]); }
            const std::string full = msg + " (matrix " + std::to_string(first) + " of the batch, status " + std::to_string(first >= 0 ? status[(size_t)first] : rc) + ")";
            if (rc == ILUPP_ERR_UNSUPPORTED) { PyErr_SetString(PyExc_NotImplementedError, full.c_str()); throw py::error_already_set(); }
            throw std::runtime_error(full);
        }
        py::list res;
        for (int32_t k = 0; k < cnt; ++k) {
            Multilevel f;
            f.h = out[(size_t)k]; f.n = N[(size_t)k];
            res.append(py::cast(std::move(f)));
        }
        return res;
    });

    // binding.cpp:200-230, bound at :281
    m.def("solve", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, py::buffer rhs, double rtol, double atol, int32_t max_iter,
                      py::object param) {
        const Csr a = borrow(data, indices, indptr, is_csr);
        py::buffer_info b = reals(rhs, "b");
        if (b.shape[0] != a.n) throw std::runtime_error("right-hand side has wrong size!");
        const ilupp_ml_params p = block_of(param);
        py::array_t<double> result(a.n);
        double *x = static_cast<double *>(result.request().ptr);
        int32_t it = 0; double rel = 0.0, res = 0.0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = ilupp_hip_solve(a.val, a.idx, a.ptr, a.n, a.row_major, static_cast<const double *>(b.ptr), b.shape[0], rtol, atol, max_iter, &p, x, &it, &rel,
                                 &res);
        }
        if (rc == ILUPP_ERR_UNSUPPORTED) { PyErr_SetString(PyExc_NotImplementedError, ilupp_hip_last_error()); throw py::error_already_set(); }
        ok(rc);
        return py::make_tuple(result, it, rel, res);
    });

    m.def("ILU0Preconditioner", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr) {
        const Csr a = borrow(data, indices, indptr, is_csr);
        return adopt<LU>(build([&](ilupp_precond **h) { return ilupp_hip_ilu0_create(a.val, a.idx, a.ptr, a.n, a.row_major, h); }));
    });
    m.def("IChol0Preconditioner", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr) {
        const Csr a = borrow(data, indices, indptr, is_csr);
        return adopt<LLT>(build([&](ilupp_precond **h) { return ilupp_hip_ichol0_create(a.val, a.idx, a.ptr, a.n, a.row_major, h); }));
    });
    m.def("ICholTPreconditioner", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t add_fill_in, double threshold) {
        const Csr a = borrow(data, indices, indptr, is_csr);
        return adopt<LLT>(build([&](ilupp_precond **h) { return ilupp_hip_icholt_create(a.val, a.idx, a.ptr, a.n, a.row_major, add_fill_in, threshold, h); }));
    });

    // stand-alone factor functions: build, hand the factors out, drop the object
    m.def("ilu0", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr) {
        const Csr a = borrow(data, indices, indptr, is_csr);
        const LU f = adopt<LU>(build([&](ilupp_precond **h) { return ilupp_hip_ilu0_create(a.val, a.idx, a.ptr, a.n, a.row_major, h); }));
        return py::tuple(egress(f));
    });
    m.def("ilut", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t fill_in, double threshold) {
        const Csr a = borrow(data, indices, indptr, is_csr);
        const ILUT f = adopt<ILUT>(build([&](ilupp_precond **h) { return ilupp_hip_ilut_create(a.val, a.idx, a.ptr, a.n, a.row_major, fill_in, threshold, h); }));
        return py::tuple(egress(f));
    });
    m.def("iluc", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t fill_in, double threshold) {
        const Csr a = borrow(data, indices, indptr, is_csr);
        const ILUC f = adopt<ILUC>(build([&](ilupp_precond **h) { return ilupp_hip_iluc_create(a.val, a.idx, a.ptr, a.n, a.row_major, fill_in, threshold, h); }));
        return py::tuple(egress(f));
    });
    m.def("ichol0", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr) -> py::object {
        const Csr a = borrow(data, indices, indptr, is_csr);
        const LLT f = adopt<LLT>(build([&](ilupp_precond **h) { return ilupp_hip_ichol0_create(a.val, a.idx, a.ptr, a.n, a.row_major, h); }));
        return py::object(egress(f)[0]);
    });
    m.def("icholt", [](py::buffer data, py::buffer indices, py::buffer indptr, bool is_csr, int32_t add_fill_in, double threshold) -> py::object {
        const Csr a = borrow(data, indices, indptr, is_csr);
        const LLT f = adopt<LLT>(build([&](ilupp_precond **h) { return ilupp_hip_icholt_create(a.val, a.idx, a.ptr, a.n, a.row_major, add_fill_in, threshold, h); }));
        return py::object(egress(f)[0]);
    });
}
