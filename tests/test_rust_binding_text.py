"""The Rust binding under rust/ cannot be compiled in this image (no cargo / rustc), so nothing would notice if it drifted away from
the C ABI.  These checks parse the TEXT of rust/chalametpir_hip_sys/src/lib.rs and rust/server_hip.rs and hold them to
include/chalamet_hip.h: same symbol set, same argument counts, same constants, same struct fields in the same order with the same
widths, every status code mapped onto a ChalametPIRError variant (reference chalametpir_common/src/error.rs:7-49)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "chalamet_hip.h")
SYS_RS = os.path.join(ROOT, "rust", "chalametpir_hip_sys", "src", "lib.rs")
SERVER_RS = os.path.join(ROOT, "rust", "server_hip.rs")


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _strip_rs(text):
    return re.sub(r"//[^\n]*", "", text)


def _split_args(arglist):
    """top-level comma split (array types such as `const uint8_t seed[32]` / `[u8; 0]` contain no commas at depth 0)"""
    arglist = arglist.strip()
    if arglist in ("", "void"):
        return []
    out, depth, cur = [], 0, ""
    for ch in arglist:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def header_functions():
    text = _strip_c(open(HEADER).read())
    text = re.sub(r"#define[^\n]*", "", text)
    fns = {}
    for m in re.finditer(r"\b(cpir_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        fns[m.group(1)] = _split_args(" ".join(m.group(2).split()))
    return fns


def rust_functions():
    text = _strip_rs(open(SYS_RS).read())
    fns = {}
    for m in re.finditer(r"pub fn (cpir_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", text, flags=re.S):
        fns[m.group(1)] = (_split_args(" ".join(m.group(2).split())), (m.group(3) or "").strip())
    return fns


def header_constants():
    raw = open(HEADER).read()
    text = _strip_c(raw)
    consts = {}
    for m in re.finditer(r"#define\s+(CPIR_[A-Z0-9_]+)\s+(\d+)u?\b", raw):
        consts[m.group(1)] = int(m.group(2))
    enum = re.search(r"typedef enum cpir_status \{(.*?)\}", text, flags=re.S).group(1)
    for m in re.finditer(r"(CPIR_[A-Z0-9_]+)\s*=\s*(\d+)", enum):
        consts[m.group(1)] = int(m.group(2))
    return consts


def rust_constants():
    text = _strip_rs(open(SYS_RS).read())
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (CPIR_[A-Z0-9_]+)\s*:\s*[a-z0-9_]+\s*=\s*(\d+)\s*;", text)}


def test_same_symbols_and_argument_counts():
    h, r = header_functions(), rust_functions()
    assert len(h) >= 50
    assert set(h) == set(r), sorted(set(h) ^ set(r))
    for name, args in h.items():
        assert len(args) == len(r[name][0]), (name, args, r[name][0])


def test_argument_and_return_widths_agree():
    """pointer-ness and integer width of every argument, and the return type, position by position"""
    def c_kind(arg):
        a = arg.replace("const ", "").strip()
        if "*" in a or "[" in a:
            return "ptr"
        for t, k in (("uint64_t", "u64"), ("uint32_t", "u32"), ("size_t", "usize"), ("uint8_t", "u8"), ("int", "c_int"), ("double", "f64")):
            if re.match(rf"{t}\b", a):
                return k
        raise AssertionError(arg)

    def rs_kind(arg):
        t = arg.split(":", 1)[1].strip() if ":" in arg else arg.strip()
        return "ptr" if t.startswith("*") else t

    h, r = header_functions(), rust_functions()
    for name, args in h.items():
        assert [c_kind(a) for a in args] == [rs_kind(a) for a in r[name][0]], name
    raw = _strip_c(open(HEADER).read())
    for name in h:
        m = re.search(rf"([A-Za-z_0-9 \*]+?)\b{name}\s*\(", raw)
        ret_c = " ".join(m.group(1).split())
        ret_rs = r[name][1]
        want = ("" if ret_c == "void" else "ptr" if "*" in ret_c else c_kind(ret_c + " x"))
        got = ("" if ret_rs == "" else "ptr" if ret_rs.startswith("*") else ret_rs)
        assert want == got, (name, ret_c, ret_rs)


def test_same_constants():
    h, r = header_constants(), rust_constants()
    status = {k for k in h if k == "CPIR_OK" or k.startswith("CPIR_ERR_")}
    assert len(status) == 18
    for k in status | {"CPIR_LWE_DIMENSION", "CPIR_SEED_BYTE_LEN", "CPIR_FILTER_PARAM_BYTE_LEN", "CPIR_SETUP_TIMING_COUNT", "CPIR_HOST_PATH_COUNT"}:
        assert k in r and r[k] == h[k], k
    for k, v in r.items():  # nothing in the binding that the header does not define
        assert h.get(k) == v, k


def test_struct_fields_agree_with_header_and_python_binding():
    from chalametpir_amd import _native

    text = _strip_c(open(HEADER).read())
    rs = _strip_rs(open(SYS_RS).read())
    width = {"uint64_t": "u64", "uint32_t": "u32"}
    for cname, pycls in (("cpir_dtc_layout", _native.DtcLayout), ("cpir_kv_db", _native.KvDb)):
        body = re.search(rf"typedef struct {cname} \{{(.*?)\}} {cname};", text, flags=re.S).group(1)
        c_fields = [(" ".join(m.group(1).split()), m.group(2)) for m in re.finditer(r"([A-Za-z_0-9 \*]+?)\b([a-z_0-9]+)\s*;", body)]
        rbody = re.search(rf"pub struct {cname} \{{(.*?)\}}", rs, flags=re.S).group(1)
        r_fields = [(m.group(2).strip(), m.group(1)) for m in re.finditer(r"pub ([a-z_0-9]+)\s*:\s*([^,\n]+)", rbody)]
        assert [n for _, n in c_fields] == [n for _, n in r_fields] == [n for n, _ in pycls._fields_], cname
        for (ct, n), (rt, _) in zip(c_fields, r_fields):
            if "*" in ct:
                assert rt.startswith("*const"), (cname, n)
            else:
                assert width[ct] == rt, (cname, n)


def test_server_shim_maps_every_status():
    h = header_constants()
    text = _strip_rs(open(SERVER_RS).read())
    body = re.search(r"fn map_status\(.*?\{(.*?)\n\}", text, flags=re.S).group(1)
    mapped = set(re.findall(r"sys::(CPIR_ERR_[A-Z0-9_]+)\s*=>", body))
    assert re.search(r"\n\s*_\s*=>\s*ChalametPIRError::", body), "no default arm"
    # every status that stands for a reference variant (error.rs:24-49) has its own arm; so do 'no device' and 'out of memory'
    for k, v in h.items():
        if k.startswith("CPIR_ERR_") and (v < 64 or k in ("CPIR_ERR_NO_DEVICE", "CPIR_ERR_OUT_OF_DEVICE_MEMORY")):
            assert k in mapped, k
    assert mapped <= set(h), mapped - set(h)
    # the variant each arm names is the one chalametpir_amd.errors gives the same code (the Python mirror of the same table)
    from chalametpir_amd.errors import VARIANTS

    for m in re.finditer(r"sys::(CPIR_ERR_[A-Z0-9_]+)\s*=>\s*ChalametPIRError::([A-Za-z0-9]+)", body):
        if h[m.group(1)] < 64:
            assert VARIANTS[h[m.group(1)]] == m.group(2), m.group(1)
    # the entry points the shim calls exist in the -sys crate with that many arguments
    r = rust_functions()
    for m in re.finditer(r"sys::(cpir_[a-z0-9_]+)\s*\(", text):
        assert m.group(1) in r, m.group(1)
