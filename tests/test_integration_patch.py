"""integration/oarfish-mi355x.patch: the reference-side binding as a diff against COMBINE-lab/oarfish v0.10.3 -- the
`mi355x` cargo feature, `src/em_gpu.rs` (extern "C" block + shim), the `boundaries()` accessor and the feature-gated
call sites (bulk.rs:155-159, :179, single_cell.rs:150).  No Rust toolchain exists in this image, so the patch is pinned
as far as the image allows: it applies to the reference checkout (when that is present), and its extern declarations
and #[repr(C)] structs agree with include/oarfish_em.h name for name, argument count for argument count, field for
field."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "oarfish-mi355x.patch")
HEADER = os.path.join(ROOT, "include", "oarfish_em.h")
REFERENCE = "/root/reference"


def _added_file(patch_text, path):
    """the text of a file the patch creates"""
    m = re.search(r"^\+\+\+ b/" + re.escape(path) + r"\n@@[^\n]*\n((?:\+[^\n]*\n)+)", patch_text, flags=re.M)
    assert m, f"{path} not created by the patch"
    return "".join(line[1:] + "\n" for line in m.group(1).splitlines())


def _header_decls():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for name, params in re.findall(r"\b(oem_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        p = params.strip()
        out[name] = 0 if p in ("", "void") else p.count(",") + 1
    return out, src


def test_patch_applies_to_the_reference_checkout():
    if not os.path.isdir(os.path.join(REFERENCE, "src")):
        pytest.skip("no reference checkout here (the GPU box has none)")
    r = subprocess.run(["git", "-C", REFERENCE, "apply", "--check", PATCH], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_patch_is_a_diff_not_a_copy():
    """three lines of context around each call site; the only whole file is the new one"""
    text = open(PATCH).read()
    new_files = re.findall(r"^--- /dev/null\n\+\+\+ b/(\S+)", text, flags=re.M)
    assert new_files == ["src/em_gpu.rs"]
    context = [l for l in text.splitlines() if l.startswith(" ")]
    assert len(context) < 80, "context lines of the reference in the patch"


def test_extern_block_matches_the_header():
    rs = _added_file(open(PATCH).read(), "src/em_gpu.rs")
    decls, _ = _header_decls()
    block = re.search(r'unsafe extern "C" \{(.*?)\n\}', rs, flags=re.S).group(1)
    fns = re.findall(r"fn (oem_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S)
    assert len(fns) >= 6
    for name, params in fns:
        assert name in decls, f"{name} is not declared by include/oarfish_em.h"
        n = len([p for p in params.split(",") if p.strip()])
        assert n == decls[name], f"{name}: {n} arguments in the Rust declaration, {decls[name]} in the header"


def test_repr_c_structs_match_the_header():
    rs = _added_file(open(PATCH).read(), "src/em_gpu.rs")
    _, hdr = _header_decls()

    def c_fields(struct):
        body = re.search(r"typedef struct\s*\{([^{}]*)\}\s*" + struct + r"\s*;", hdr, flags=re.S).group(1)
        return [re.sub(r"\[.*", "", f.strip().split()[-1]) for f in body.split(";") if f.strip()]

    def rs_fields(struct):
        body = re.search(r"pub struct " + struct + r"\s*\{(.*?)\n\}", rs, flags=re.S).group(1)
        return re.findall(r"pub ([a-z_0-9]+)\s*:", body)

    assert rs_fields("OemRunInfo") == c_fields("oem_run_info")
    assert rs_fields("OemStoreOpts") == c_fields("oem_store_opts")
    assert "oem_abi_version() } == %s" % re.search(r"#define OEM_ABI_VERSION (\d+)", open(HEADER).read()).group(1) in rs
