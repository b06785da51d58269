"""Validated, frozen pipeline settings (reference torch_darktable/pipeline/config.py).

Same field names, defaults, ranges and JSON form as the reference so its camera JSON files load
unchanged; written for Python 3.10 (the reference uses PEP 695 generics, config.py:54)."""

from __future__ import annotations

from enum import Enum
from pathlib import Path
from typing import Annotated, Literal, get_args, get_origin

from pydantic import BaseModel, GetCoreSchemaHandler
from pydantic_core import core_schema


class Validator:
    """Annotation marker carrying a UI description and a pydantic core schema."""

    description: str


class _Ranged(Validator):
    cast = float

    def __init__(self, range: tuple, description: str):
        self.range = range
        self.description = description

    def __get_pydantic_core_schema__(self, _source_type, _handler: GetCoreSchemaHandler):
        lo, hi = self.range
        cast = self.cast

        def validate(v):
            v = cast(v)
            if not (lo <= v <= hi):
                raise ValueError(f'{v} not in [{lo}, {hi}]')
            return v

        return core_schema.no_info_plain_validator_function(validate)


class Float(_Ranged):
    cast = float


class Int(_Ranged):
    cast = int

    def __init__(self, range: tuple[int, int], description: str, step: int | None = None):
        super().__init__(range, description)
        self.step = step


class Bool(Validator):
    def __init__(self, description: str):
        self.description = description

    def __get_pydantic_core_schema__(self, _source_type, _handler: GetCoreSchemaHandler):
        return core_schema.no_info_plain_validator_function(bool)


class EnumValidator(Validator):
    """Enum field stored by NAME in JSON; also accepts {key: name} dicts (per-camera transforms)."""

    def __init__(self, enum_type: type[Enum], description: str):
        self.enum_type = enum_type
        self.description = description

    def __get_pydantic_core_schema__(self, _source_type, _handler: GetCoreSchemaHandler):
        enum_type = self.enum_type

        def by_name(name: str):
            try:
                return enum_type[name]
            except KeyError:  # surfaces as a pydantic ValidationError (the reference lets the KeyError escape)
                raise ValueError(f'{name!r} is not a {enum_type.__name__} ({[m.name for m in enum_type]})') from None

        def validate(v):
            if isinstance(v, enum_type):
                return v
            if isinstance(v, str):
                return by_name(v)
            if isinstance(v, dict):
                return {k: by_name(x) if isinstance(x, str) else x for k, x in v.items()}
            raise ValueError(f'{v} is not a {enum_type.__name__}')

        def serialize(v):
            return {k: x.name for k, x in v.items()} if isinstance(v, dict) else v.name

        return core_schema.no_info_plain_validator_function(
            validate, serialization=core_schema.plain_serializer_function_ser_schema(serialize, when_used='always'))


def get_validator(model: type[BaseModel], field_name: str) -> Validator | None:
    """The Validator attached to a field's Annotated[...] type, if any."""
    ann = model.__annotations__.get(field_name)
    if isinstance(ann, str):  # `from __future__ import annotations`: resolve lazily
        import typing

        ann = typing.get_type_hints(model, include_extras=True).get(field_name)
    if ann is not None and get_origin(ann) is Annotated:
        for meta in get_args(ann)[1:]:
            if isinstance(meta, Validator):
                return meta
    return None


class ToneMapper(Enum):
    linear = 0
    reinhard = 1
    aces = 2
    adaptive_aces = 3


class Debayer(Enum):
    bilinear = 0
    ppg = 1
    rcd = 2


def clamp(x, lower, upper):
    return min(max(x, lower), upper)


class ImageProcessingSettings(BaseModel, frozen=True):
    type: Literal['image_processing_settings'] = 'image_processing_settings'

    tone_gamma: Annotated[float, Float(range=(0.1, 5.0), description='Gamma')] = 0.75
    tone_intensity: Annotated[float, Float(range=(-1.0, 5.0), description='Intensity')] = 2.0
    light_adapt: Annotated[float, Float(range=(0.0, 1.0), description='Light adaptation')] = 1.0
    vibrance: Annotated[float, Float(range=(-1.0, 1.0), description='Vibrance')] = 0.0

    # exponential moving average of bounds / metrics across calls (1 = per-call statistics)
    moving_average: Annotated[float, Float(range=(0.0, 1.0), description='Tonemap moving average')] = 0.02

    debayer: Annotated[Debayer, EnumValidator(Debayer, description='Debayer algorithm')] = Debayer.rcd
    ppg_median_threshold: float = 0.0

    postprocess: Annotated[bool, Bool(description='Postprocess debayer')] = False
    green_eq_threshold: float = 0.04
    color_smoothing_passes: int = 3

    enable_bilateral: Annotated[bool, Bool(description='Enable bilateral constrast enhancement')] = False
    bilateral: Annotated[float, Float(range=(0.0, 1.0), description='Bilateral constrast enhancement amount')] = 0.4
    bil_sigma_spatial: float = 2.0
    bil_sigma_luminance: float = 0.2

    enable_denoise: Annotated[bool, Bool(description='Enable denoise')] = True
    denoise: Annotated[float, Float(range=(0.0, 1.0), description='Denoise amount')] = 0.075

    tone_mapping: Annotated[ToneMapper, EnumValidator(ToneMapper, description='Tonemapping algorithm')] = ToneMapper.reinhard

    resize_width: Annotated[int, Int(range=(0, 4096), description='Resize width')] = 0

    def save_json(self, path: Path) -> None:
        Path(path).write_text(self.model_dump_json(indent=2))

    @classmethod
    def load_json(cls, path: Path) -> 'ImageProcessingSettings':
        return cls.model_validate_json(Path(path).read_text())
