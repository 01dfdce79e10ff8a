"""Input-tuple contract of every careless model component.

Mirror of `careless/models/base.py:6-121` (reference): same `input_index`, same accessor names, same error behaviour.
Inputs may be numpy arrays or torch tensors; ids are int64 and data float32, all 2-D, as the reference formatters emit
them (`careless/io/formatter.py:382-394`).
"""
from __future__ import annotations


class BaseModel:
    """Base class for all models.  Encodes accessors for the standard format inputs (reference `BaseModel`)."""

    input_index = {
        "refl_id": 0,
        "image_id": 1,
        "file_id": 2,
        "metadata": 3,
        "intensities": 4,
        "uncertainties": 5,
        "wavelength": 6,
        "harmonic_id": 7,
    }

    def call(self, inputs):
        raise NotImplementedError(
            "All Scaler classes must implement a call method which accepts inputs defined by this class.")

    def __call__(self, inputs, *args, **kwargs):
        return self.call(inputs, *args, **kwargs)

    @staticmethod
    def is_laue(inputs) -> bool:
        """Laue data carry wavelength and harmonic_id (reference base.py:39-47)."""
        return len(inputs) >= BaseModel.get_index_by_name("harmonic_id") + 1

    @staticmethod
    def get_name_by_index(index: int) -> str:
        for k, v in BaseModel.input_index.items():
            if v == index:
                return k
        raise ValueError(
            f"index, {index}, not a valid index. Valid indices are {BaseModel.input_index.values()}.")

    @staticmethod
    def get_index_by_name(name):
        if name not in BaseModel.input_index:
            raise ValueError(f"name, {name}, not a valid key. Valid keys are {BaseModel.input_index.keys()}.")
        return BaseModel.input_index[name]

    @staticmethod
    def get_input_by_name(inputs, name):
        if name not in BaseModel.input_index:
            raise ValueError(f"name, {name}, not a valid key. Valid keys are {BaseModel.input_index.keys()}.")
        idx = BaseModel.input_index[name]
        try:
            datum = inputs[idx]
        except Exception:
            raise ValueError(
                f"Attempting to gather {name} data from input tensors, {inputs}, with length {len(inputs)} failed.")
        if datum.shape[0] == 1 and datum.ndim > 2:        # a leading batch axis of 1 is squeezed (base.py:79-80)
            datum = datum[0]
        return datum

    @staticmethod
    def get_refl_id(inputs):
        return BaseModel.get_input_by_name(inputs, "refl_id")

    @staticmethod
    def get_file_id(inputs):
        return BaseModel.get_input_by_name(inputs, "file_id")

    @staticmethod
    def get_image_id(inputs):
        return BaseModel.get_input_by_name(inputs, "image_id")

    @staticmethod
    def get_metadata(inputs):
        return BaseModel.get_input_by_name(inputs, "metadata")

    @staticmethod
    def get_intensities(inputs):
        return BaseModel.get_input_by_name(inputs, "intensities")

    @staticmethod
    def get_uncertainties(inputs):
        return BaseModel.get_input_by_name(inputs, "uncertainties")

    @staticmethod
    def get_wavelength(inputs):
        return BaseModel.get_input_by_name(inputs, "wavelength")

    @staticmethod
    def get_harmonic_id(inputs):
        return BaseModel.get_input_by_name(inputs, "harmonic_id")
