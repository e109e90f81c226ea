"""JSON settings loader (mirrors reference cli/json_loader.py:5-18)."""
import json


def load_input_json(file_path: str) -> dict:
    """Return the settings dictionary stored in ``file_path``."""
    with open(file_path) as handle:
        return json.load(handle)
