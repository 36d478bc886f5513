"""Config base class with the JSON / YAML round trip of the reference's ``BaseConfig``
(src/lbm/config.py:13-141): same method names, same ``name`` field handling."""
import json
import os
import warnings
from dataclasses import asdict, field
from typing import Any, Dict, Union

import yaml
from pydantic.dataclasses import dataclass


@dataclass
class BaseConfig:
    name: str = field(init=False)

    def __post_init__(self):
        self.name = self.__class__.__name__

    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> "BaseConfig":
        return cls(**config_dict)

    @classmethod
    def _dict_from_json(cls, json_path: Union[str, os.PathLike]) -> Dict[str, Any]:
        try:
            with open(json_path) as f:
                try:
                    return json.load(f)
                except (TypeError, json.JSONDecodeError) as e:
                    raise TypeError(f"File {json_path} not loadable. Maybe not json ? \n"
                                    f"Catch Exception {type(e)} with message: " + str(e)) from e
        except FileNotFoundError:
            raise FileNotFoundError(f"Config file not found. Please check path '{json_path}'")

    @classmethod
    def _checked(cls, config_dict):
        config_name = config_dict.pop("name")
        if cls.__name__ != config_name:
            warnings.warn(f"You are trying to load a `{cls.__name__}` while a `{config_name}` is given.")
        return cls.from_dict(config_dict)

    @classmethod
    def from_json(cls, json_path: str) -> "BaseConfig":
        return cls._checked(cls._dict_from_json(json_path))

    def to_dict(self) -> dict:
        return asdict(self)

    def to_json_string(self):
        return json.dumps(self.to_dict())

    def save_json(self, file_path: str):
        with open(os.path.join(file_path), "w", encoding="utf-8") as fp:
            fp.write(self.to_json_string())

    def save_yaml(self, file_path: str):
        with open(os.path.join(file_path), "w", encoding="utf-8") as fp:
            yaml.dump(self.to_dict(), fp)

    @classmethod
    def from_yaml(cls, yaml_path: str) -> "BaseConfig":
        with open(yaml_path, "r") as f:
            try:
                config_dict = yaml.safe_load(f)
            except yaml.YAMLError as e:
                raise yaml.YAMLError(f"File {yaml_path} not loadable. Maybe not yaml ? \n"
                                     f"Catch Exception {type(e)} with message: " + str(e)) from e
        return cls._checked(config_dict)
