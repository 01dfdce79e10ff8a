"""The positional contract of the `inputs` tuple shared by every model component.

Public surface = the reference's `BaseModel` (careless/models/base.py:6-121): the `input_index` table, `get_<column>(inputs)`
accessors, `get_input_by_name`, `get_index_by_name`, `get_name_by_index`, `is_laue`, and `__call__` forwarding to `call`.
Columns are numpy arrays or torch tensors, all 2-D: ids int64, data float32 (careless/io/formatter.py:382-394).
The accessors are generated from the table below rather than written out one by one.
"""
from __future__ import annotations

_COLUMNS = ("refl_id", "image_id", "file_id", "metadata", "intensities", "uncertainties", "wavelength", "harmonic_id")


def _bad_name(name):
    return ValueError(f"name, {name}, not a valid key. Valid keys are {BaseModel.input_index.keys()}.")


class BaseModel:
    """Common base of priors, likelihoods, scalers and the merging model: knows where each column of `inputs` lives."""

    input_index = {name: position for position, name in enumerate(_COLUMNS)}

    # -- protocol ---------------------------------------------------------------------------------------------
    def call(self, inputs):
        raise NotImplementedError("All Scaler classes must implement a call method which accepts inputs defined by this class.")

    def __call__(self, inputs, *args, **kwargs):
        return self.call(inputs, *args, **kwargs)

    # -- table lookups ----------------------------------------------------------------------------------------
    @staticmethod
    def get_index_by_name(name):
        try:
            return BaseModel.input_index[name]
        except KeyError:
            raise _bad_name(name) from None

    @staticmethod
    def get_name_by_index(index: int) -> str:
        if isinstance(index, int) and 0 <= index < len(_COLUMNS):
            return _COLUMNS[index]
        raise ValueError(f"index, {index}, not a valid index. Valid indices are {BaseModel.input_index.values()}.")

    @staticmethod
    def is_laue(inputs) -> bool:
        """Polychromatic inputs are the ones that reach as far as the harmonic_id column (reference base.py:39-47)."""
        return len(inputs) > BaseModel.input_index["harmonic_id"]

    @staticmethod
    def get_input_by_name(inputs, name):
        position = BaseModel.get_index_by_name(name)
        if position >= len(inputs):
            raise ValueError(f"Attempting to gather {name} data from input tensors, {inputs}, with length {len(inputs)} failed.")
        column = inputs[position]
        # a data set delivered as one batch carries a leading axis of length 1: drop it (reference base.py:79-80)
        return column[0] if (column.ndim > 2 and column.shape[0] == 1) else column


def _make_getter(column_name):
    def getter(inputs):
        return BaseModel.get_input_by_name(inputs, column_name)
    getter.__name__ = f"get_{column_name}"
    getter.__doc__ = f"The `{column_name}` column of `inputs`."
    return staticmethod(getter)


for _name in _COLUMNS:
    setattr(BaseModel, f"get_{_name}", _make_getter(_name))
del _name
