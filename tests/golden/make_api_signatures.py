"""Record the public API of the reference's plugin surface WITHOUT importing it: the files are parsed with `ast` only
(several of them cannot be imported here at all -- they load the CUDA extension or tinycudann at import time).

    python tests/golden/make_api_signatures.py        # build container only (/root/reference)

Writes tests/golden/api_signatures.json: for every module below, its module-level functions and its classes with their
bases, their methods (parameter names in order, which have defaults, *args / **kwargs, decorators such as property /
abstractmethod / staticmethod) and their annotated dataclass-style fields.  tests/test_api_surface.py holds the mirror
package (gf-nerf_amd/gfnerf/) to it.  The JSON is data about names; no reference source text is stored."""
import ast
import json
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = [
    "gfnerf/gf_pipeline.py", "gfnerf/nerfacto.py", "gfnerf/nerfacto_field.py", "gfnerf/perssampler.py", "gfnerf/hash_3d_anchored.py",
    "gfnerf/mlp.py", "gfnerf/config.py", "nerfstudio/models/base_model.py", "nerfstudio/fields/base_field.py",
    "nerfstudio/plugins/types.py", "nerfstudio/plugins/registry.py", "nerfstudio/pipelines/base_pipeline.py",
]


def _name(node):
    if isinstance(node, ast.Name):
        return node.id
    if isinstance(node, ast.Attribute):
        return _name(node.value) + "." + node.attr
    if isinstance(node, ast.Call):
        return _name(node.func)
    if isinstance(node, ast.Subscript):
        return _name(node.value)
    return type(node).__name__


def _func(fn):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    n_def = len(a.defaults)
    return dict(params=pos, defaults=pos[len(pos) - n_def:] if n_def else [], kwonly=[x.arg for x in a.kwonlyargs],
                vararg=a.vararg.arg if a.vararg else None, kwarg=a.kwarg.arg if a.kwarg else None,
                decorators=[_name(d) for d in fn.decorator_list], line=fn.lineno)


def parse(path):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    mod = dict(functions={}, classes={}, assignments=[])
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            mod["functions"][node.name] = _func(node)
        elif isinstance(node, ast.ClassDef):
            cls = dict(bases=[_name(b) for b in node.bases], decorators=[_name(d) for d in node.decorator_list], methods={}, fields=[], line=node.lineno)
            for item in node.body:
                if isinstance(item, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    cls["methods"][item.name] = _func(item)
                elif isinstance(item, ast.AnnAssign) and isinstance(item.target, ast.Name):
                    cls["fields"].append(item.target.id)
            mod["classes"][node.name] = cls
        elif isinstance(node, ast.Assign):
            mod["assignments"] += [t.id for t in node.targets if isinstance(t, ast.Name)]
        elif isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
            mod["assignments"].append(node.target.id)
    return mod


BINDINGS = "gfnerf/bindings/hashanchored/bindings.cpp"


def torchscript_classes(path):
    """Names only: the TorchScript classes TORCH_LIBRARY(my_classes, ...) registers and the method names each `.def("name", ...)` exposes
    (bindings.cpp:343-399) -> {class: [method names in order]}."""
    import re

    text = open(os.path.join(REF, path)).read()
    text = re.sub(r"//[^\n]*", "", text[text.index("TORCH_LIBRARY("):])
    out = {}
    for m in re.finditer(r'm\.class_<\w+>\("(\w+)"\)(.*?);', text, re.S):
        out[m.group(1)] = re.findall(r'\.def\("(\w+)"', m.group(2))
    return out


def main():
    assert os.path.isdir(REF), "reference tree not mounted; the signature file can only be regenerated in the build container"
    out = {p: parse(p) for p in FILES}
    out[BINDINGS] = dict(torchscript_classes=torchscript_classes(BINDINGS))
    path = os.path.join(HERE, "api_signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    n_cls = sum(len(m.get("classes", {})) for m in out.values())
    n_meth = sum(len(c["methods"]) for m in out.values() for c in m.get("classes", {}).values())
    print(f"wrote {path}: {len(out)} modules, {n_cls} classes, {n_meth} methods")


if __name__ == "__main__":
    sys.exit(main())
