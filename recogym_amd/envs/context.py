"""Context of an observation: (time, user) — reference: recogym/envs/context.py."""


class Context:
    def time(self):
        raise NotImplementedError

    def user(self):
        raise NotImplementedError


class DefaultContext(Context):
    # event_index: the per-user event number the addressed policy draws are keyed by; equals time() with the default
    # time generator and is set by the env when a NormalTimeGenerator makes time() a float (not in the reference)
    __slots__ = ('current_time', 'current_user_id', 'event_index')

    def __init__(self, current_time, current_user_id, event_index=None):
        self.current_time = current_time
        self.current_user_id = current_user_id
        self.event_index = event_index

    def draw_key(self):
        """(user, t) the policy's addressed draws use."""
        return self.current_user_id, (self.current_time if self.event_index is None else self.event_index)

    def time(self):
        return self.current_time

    def user(self):
        return self.current_user_id
