"""On-disk helpers with the names and file formats of PokerRL/util/file_util.py: `<name>.json` / `<name>.js`
(PokerViz reads `const data=<json>`) / `<name>.pkl` (highest pickle protocol). Files written by either package load in the other."""
import json
import os
import pickle
from pathlib import Path


def create_dir_if_not_exist(path):
    Path(path).mkdir(parents=True, exist_ok=True)


def get_all_files_in_dir(_dir):
    return [e.name for e in os.scandir(_dir) if e.is_file()]


def get_all_dirs_in_dir(_dir):
    return [e.name for e in os.scandir(_dir) if e.is_dir()]


def get_file_name_without_ending_and_path_from_path(path):
    return Path(path).stem


def _write_text(_dir, file_name, suffix, text):
    create_dir_if_not_exist(_dir)
    Path(_dir, str(file_name) + suffix).write_text(text)


def write_dict_to_file_json(_dir, file_name, dictionary):
    _write_text(_dir, file_name, ".json", json.dumps(dictionary))


def write_dict_to_file_js(_dir, file_name, dictionary):
    _write_text(_dir, file_name, ".js", "const data=" + json.dumps(dictionary))


def do_pickle(obj, path, file_name):
    create_dir_if_not_exist(path)
    with open(Path(path, str(file_name) + ".pkl"), "wb") as f:
        pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)


def load_pickle(path, file_name=None):
    """`load_pickle(dir, name)` or `load_pickle(full_path)`"""
    with open(path if file_name is None else Path(path, str(file_name) + ".pkl"), "rb") as f:
        return pickle.load(f)
