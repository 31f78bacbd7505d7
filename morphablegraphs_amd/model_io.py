"""Readers for the on-disk formats the reference writes: motion-primitive JSON -- legacy v1/v2 and the v3
``sspm/tspm/gmm`` layout (reference motion_model/motion_model_constructor.py:431-524) -- and the graph zip
(reference utilities/zip_io.py:37-233): ``graph_definition.json`` plus, per elementary action, a directory
``elementary_action_models/elementary_action_<action>/`` with ``<action>_<primitive>_quaternion_mm.json``,
``meta_information.json`` and optionally ``<...>_quaternion_cluster_tree.json`` (format >= 4).
``read_graph_zip`` returns the same nested dict as ZipReader.get_graph_data (keys derived the same way), so
HipMotionStateGraph.build_from_graph_data reads like MotionStateGraphLoader._build_from_zip_file."""
import json
import zipfile

from .motion_primitive_wrapper import mgrd_json_to_legacy

MORPHABLE_MODEL_FILE_ENDING = "mm.json"          # zip_io.py:37-49
MM_TYPE = "quaternion"
ELEMENTARY_ACTION_DIRECTORY = "elementary_action_models"
GRAPH_DEFINITION_FILE = "graph_definition.json"
SKELETON_JSON_FILE = "skeleton.json"
SKELETON_JSON_KEY = "skeleton"
MM_SUFFIX = "_" + MM_TYPE + "_" + MORPHABLE_MODEL_FILE_ENDING


def primitive_dict_from_json(data):
    if "spatial_coeffs" in data:
        raise ValueError("static motion primitive files have no statistical model")
    return mgrd_json_to_legacy(data) if "tspm" in data else data


def load_primitive_file(path):
    with open(path, "r") as f:
        return primitive_dict_from_json(json.load(f))


def _action_key(structure_key):
    """zip_io.py:174: ``elementary_action_walk`` -> ``walk``; a plain directory name is kept."""
    parts = structure_key.split("_")
    return parts[2] if len(parts) > 2 else structure_key


def _primitive_key(motion_primitive_name):
    """zip_io.py:186-190: ``walk_leftStance_quaternion`` -> (``leftStance``, ``walk_leftStance``);
    a file without the action prefix keeps its whole stem."""
    stem = motion_primitive_name[:-(len(MM_TYPE) + 1)]
    head = stem.split("_")[0]
    key = stem[len(head) + 1:] if "_" in stem else stem
    return key, stem


def read_graph_zip(path):
    """ZipReader.get_graph_data (zip_io.py:65-95): {"subgraphs": {action: {"name", "info"?, "nodes": {primitive:
    {"name", "mm", "stats"?, "space_partition_json"?}}}}, "transitions"?, "startNode"?, "skeleton"?, ...}."""
    with zipfile.ZipFile(path, "r") as z:
        names = z.namelist()
        data = json.loads(z.read(GRAPH_DEFINITION_FILE).decode("utf-8")) if GRAPH_DEFINITION_FILE in names else {}
        version = float(data.get("formatVersion", 1.0))
        if SKELETON_JSON_FILE in names:
            data[SKELETON_JSON_KEY] = json.loads(z.read(SKELETON_JSON_FILE).decode("utf-8"))
        subgraphs = {}
        for name in names:
            parts = name.split("/")
            if not name.endswith(MORPHABLE_MODEL_FILE_ENDING):
                continue
            # the layout follows formatVersion, as ZipReader's path getters do (zip_io.py:215-233): >= 2.0 keeps the
            # actions under elementary_action_models/, 1.x at the top level
            if version >= 2.0:
                if len(parts) != 3 or parts[0] != ELEMENTARY_ACTION_DIRECTORY:
                    continue
                structure_key, file_name, prefix = parts[1], parts[2], parts[0] + "/" + parts[1] + "/"
            else:
                if len(parts) != 2:
                    continue
                structure_key, file_name, prefix = parts[0], parts[1], parts[0] + "/"
            action = _action_key(structure_key)
            group = subgraphs.setdefault(action, {"name": action, "nodes": {}})
            meta = prefix + "meta_information.json"
            if "info" not in group and meta in names:
                group["info"] = json.loads(z.read(meta).decode("utf-8"))
            motion_primitive_name = file_name[:-(len(MORPHABLE_MODEL_FILE_ENDING) + 1)]
            key, stem = _primitive_key(motion_primitive_name)
            node = {"name": stem, "mm": json.loads(z.read(name).decode("utf-8"))}
            # the reference looks the statistics up at '<structure_key>/<stem>.stats' WITHOUT the elementary_action_models
            # prefix, whatever the version (zip_io.py:196); the file next to the model is taken as a fallback
            for stats in (structure_key + "/" + stem + ".stats", prefix + stem + ".stats"):
                if stats in names:
                    node["stats"] = json.loads(z.read(stats).decode("utf-8"))
                    break
            tree = prefix + motion_primitive_name + "_cluster_tree.json"
            if version >= 4.0 and tree in names:
                node["space_partition_json"] = json.loads(z.read(tree).decode("utf-8"))
            group["nodes"][key] = node
        data["subgraphs"] = subgraphs
    return data


def load_graph_zip(path):
    """{(action, primitive): legacy dict} for every statistical primitive in a graph zip."""
    out = {}
    for action, group in read_graph_zip(path)["subgraphs"].items():
        for key, node in group["nodes"].items():
            if "spatial_coeffs" in node["mm"]:
                continue
            d = primitive_dict_from_json(node["mm"])
            d.setdefault("name", node["name"])
            out[(action, key)] = d
    return out
