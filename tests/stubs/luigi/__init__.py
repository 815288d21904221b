"""Test stub of the slice of luigi the Voxels task uses: typed parameters with defaults, tasks
constructed from keyword arguments, ``get_task_family``."""

_NO_DEFAULT = object()


class Parameter:
    def __init__(self, default=_NO_DEFAULT, description=None):
        self.default = default
        self.name = None

    def __set_name__(self, owner, name):
        self.name = name

    def normalize(self, value):
        return value

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        return obj.__dict__[self.name]

    def __set__(self, obj, value):
        obj.__dict__[self.name] = value


class FloatParameter(Parameter):
    def normalize(self, value):
        return float(value)


class BoolParameter(Parameter):
    def normalize(self, value):
        return bool(value)


class DictParameter(Parameter):
    pass


class ListParameter(Parameter):
    def normalize(self, value):
        return tuple(value)  # luigi hands lists back as tuples


class TaskParameter(Parameter):
    pass


class MissingParameterException(Exception):
    pass


class Task:
    def __init__(self, **kwargs):
        params = {}
        for klass in reversed(type(self).__mro__):
            for name, attr in vars(klass).items():
                if isinstance(attr, Parameter):
                    params[name] = attr
        for name, par in params.items():
            if name in kwargs:
                value = kwargs.pop(name)
            elif par.default is not _NO_DEFAULT:
                value = par.default
            else:
                raise MissingParameterException(f"{type(self).__name__}: no value for '{name}'")
            setattr(self, name, None if value is None else par.normalize(value))
        if kwargs:
            raise TypeError(f"unknown parameters {sorted(kwargs)}")

    @classmethod
    def get_task_family(cls):
        return cls.__name__

    @classmethod
    def get_params(cls):
        out = []
        for klass in reversed(cls.__mro__):
            for name, attr in vars(klass).items():
                if isinstance(attr, Parameter):
                    out.append((name, attr))
        return out
