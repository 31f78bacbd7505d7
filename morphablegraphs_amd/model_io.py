"""Readers for the on-disk motion-primitive formats the reference writes
(motion_model_constructor.py:431-524) -- legacy v1/v2 JSON and the v3 ``sspm/tspm/gmm`` layout -- and for the
``elementary_action_models/<action>/<name>_quaternion_mm.json`` members of a graph zip
(utilities/zip_io.py:37-233).  Returns the legacy dict HipMotionPrimitive consumes."""
import json
import zipfile

from .motion_primitive_wrapper import mgrd_json_to_legacy

MM_SUFFIX = "_quaternion_mm.json"


def primitive_dict_from_json(data):
    if "spatial_coeffs" in data:
        raise ValueError("static motion primitive files have no statistical model")
    return mgrd_json_to_legacy(data) if "tspm" in data else data


def load_primitive_file(path):
    with open(path, "r") as f:
        return primitive_dict_from_json(json.load(f))


def load_graph_zip(path):
    """{(action, primitive_name): legacy dict} for every statistical primitive in a graph zip."""
    out = {}
    with zipfile.ZipFile(path) as z:
        for name in z.namelist():
            if not name.endswith(MM_SUFFIX):
                continue
            parts = name.split("/")
            action = parts[-2] if len(parts) >= 2 else ""
            prim_name = parts[-1][: -len(MM_SUFFIX)]
            data = json.loads(z.read(name).decode("utf-8"))
            if "spatial_coeffs" in data:
                continue
            d = primitive_dict_from_json(data)
            d.setdefault("name", prim_name)
            out[(action, prim_name)] = d
    return out
