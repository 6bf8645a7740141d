"""save_dict_to_json (lib/param_dict.py:18-22) -- config dump used by the experiment's setup_log."""
import json


def save_dict_to_json(dict_to_save, json_file):
    with open(json_file, 'w') as f:
        json.dump({k: (v if isinstance(v, (int, float, str, bool, list, dict, type(None))) else str(v))
                   for k, v in dict_to_save.items()}, f, indent=4, sort_keys=True)
