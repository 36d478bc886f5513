"""``BaseConfig``: the pydantic dataclass every ``*Config`` of the package derives from.

Contract (src/lbm/config.py:13-141 of the reference, whose YAML / JSON files must keep loading):
  * a ``name`` field that is not an ``__init__`` argument and always equals the class name; it is written by
    ``to_dict`` and therefore present in saved files;
  * ``from_dict(d)`` = ``cls(**d)`` (pydantic validates); ``from_json(path)`` / ``from_yaml(path)`` read a mapping,
    take ``name`` out of it, warn when it names another class, and build the instance from the rest;
  * ``to_dict`` / ``to_json_string`` / ``save_json`` / ``save_yaml`` are the inverse;
  * a missing JSON file is a ``FileNotFoundError``, undecodable JSON a ``TypeError``, undecodable YAML a
    ``yaml.YAMLError``.
"""
import dataclasses
import json
import os
import warnings
from typing import Any, Callable, Dict, Union

import yaml
from pydantic.dataclasses import dataclass

PathLike = Union[str, os.PathLike]


def _parse_json(text: str, path: PathLike) -> Any:
    try:
        return json.loads(text)
    except (TypeError, json.JSONDecodeError) as err:
        raise TypeError(f"{path} is not valid JSON ({type(err).__name__}: {err})") from err


def _parse_yaml(text: str, path: PathLike) -> Any:
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError as err:
        raise yaml.YAMLError(f"{path} is not valid YAML ({type(err).__name__}: {err})") from err


@dataclass
class BaseConfig:
    name: str = dataclasses.field(init=False)

    def __post_init__(self):
        self.name = type(self).__name__

    # ---- reading ------------------------------------------------------------------------------------------
    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> "BaseConfig":
        return cls(**config_dict)

    @classmethod
    def _from_file(cls, path: PathLike, parse: Callable[[str, PathLike], Any]) -> "BaseConfig":
        if not os.path.exists(path):
            raise FileNotFoundError(f"config file '{path}' does not exist")
        with open(path, "r", encoding="utf-8") as fh:
            fields = dict(parse(fh.read(), path))
        stored = fields.pop("name")
        if stored != cls.__name__:
            warnings.warn(f"'{path}' was saved from a `{stored}`; loading it as a `{cls.__name__}`")
        return cls.from_dict(fields)

    @classmethod
    def _dict_from_json(cls, json_path: PathLike) -> Dict[str, Any]:
        if not os.path.exists(json_path):
            raise FileNotFoundError(f"config file '{json_path}' does not exist")
        with open(json_path, "r", encoding="utf-8") as fh:
            return _parse_json(fh.read(), json_path)

    @classmethod
    def from_json(cls, json_path: PathLike) -> "BaseConfig":
        return cls._from_file(json_path, _parse_json)

    @classmethod
    def from_yaml(cls, yaml_path: PathLike) -> "BaseConfig":
        return cls._from_file(yaml_path, _parse_yaml)

    # ---- writing ------------------------------------------------------------------------------------------
    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    def to_json_string(self) -> str:
        return json.dumps(self.to_dict())

    def save_json(self, file_path: PathLike) -> None:
        with open(file_path, "w", encoding="utf-8") as fh:
            fh.write(self.to_json_string())

    def save_yaml(self, file_path: PathLike) -> None:
        with open(file_path, "w", encoding="utf-8") as fh:
            yaml.dump(self.to_dict(), fh)
