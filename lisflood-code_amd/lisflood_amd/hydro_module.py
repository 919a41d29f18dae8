"""HydroModule base -- same shape as the reference's hydrological_modules/__init__.py:49-76, so the
engine's modules can sit next to (or replace) the reference's in `lisflood.hydrological_modules`."""


class HydroModule(object):
    input_files_keys = None
    module_name = None

    def initial(self):
        pass

    def dynamic(self, *args, **kwargs):
        raise NotImplementedError

    @classmethod
    def check_input_files(cls, option):
        """The reference validates binding keys against the XML settings here
        (hydrological_modules/__init__.py:58-76); settings/IO are outside this engine's scope, so the
        classmethod only reports which keys the module would need."""
        keys = []
        for k, v in (cls.input_files_keys or {}).items():
            if k == "all" or option.get(k):
                keys += list(v)
        return keys
