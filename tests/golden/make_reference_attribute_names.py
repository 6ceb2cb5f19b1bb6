#!/usr/bin/env python3
"""Generates tests/golden/reference_attribute_names.json: for every class of the reference's latticenet_py/lattice/lattice_modules.py
and models.py, the names it assigns on `self` anywhere in its body (`self.<name> = ...`) and its base classes.  A state_dict key is
a path of exactly these names (plus ModuleList indices and torch's own parameter names), so the list pins the checkpoint key space
to the reference's source without executing it (it needs the compiled `latticenet` extension and easypbr).  Only NAMES are
extracted (ast); no source text is kept.  Run in the build container: python tests/golden/make_reference_attribute_names.py"""
import ast
import json
import os

REF = "/root/reference/latticenet_py/lattice"
out = {}
for fn in ("lattice_modules.py", "models.py"):
    tree = ast.parse(open(os.path.join(REF, fn)).read())
    for node in tree.body:
        if not isinstance(node, ast.ClassDef):
            continue
        names = set()
        for sub in ast.walk(node):
            targets = []
            if isinstance(sub, ast.Assign):
                targets = sub.targets
            elif isinstance(sub, (ast.AugAssign, ast.AnnAssign)):
                targets = [sub.target]
            for t in targets:
                for el in (t.elts if isinstance(t, (ast.Tuple, ast.List)) else [t]):
                    if isinstance(el, ast.Attribute) and isinstance(el.value, ast.Name) and el.value.id == "self":
                        names.add(el.attr)
        bases = [b.attr if isinstance(b, ast.Attribute) else getattr(b, "id", "?") for b in node.bases]
        out[node.name] = {"file": fn, "bases": bases, "self_attributes": sorted(names)}
# classes made by `X = weight_norm_wrapper(Base, ...)` at module level (lattice_modules.py and utils/utils.py): the wrapper hands the
# parameter `name` (its default is read from the function's signature) to torch.nn.utils.weight_norm, which replaces it by
# `<name>_g` / `<name>_v`
utils_tree = ast.parse(open(os.path.join(REF, "..", "utils", "utils.py")).read())
wn_default = "weight"
for node in utils_tree.body:
    if isinstance(node, ast.FunctionDef) and node.name == "weight_norm_wrapper":
        args = node.args.args
        defaults = dict(zip([a.arg for a in args[len(args) - len(node.args.defaults):]], node.args.defaults))
        wn_default = defaults["name"].value
for fn, tree in (("lattice_modules.py", ast.parse(open(os.path.join(REF, "lattice_modules.py")).read())), ("../utils/utils.py", utils_tree)):
    for node in tree.body:
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            f = node.value.func
            fname = f.attr if isinstance(f, ast.Attribute) else getattr(f, "id", "")
            if fname == "weight_norm_wrapper" and node.value.args:
                b = node.value.args[0]
                base = b.attr if isinstance(b, ast.Attribute) else getattr(b, "id", "?")
                name = next((k.value.value for k in node.value.keywords if k.arg == "name"), wn_default)
                out[node.targets[0].id] = {"file": fn, "weight_norm_of": base, "weight_norm_name": name}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_attribute_names.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(f"{len(out)} classes -> {path}")
