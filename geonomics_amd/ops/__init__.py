"""Operations that run between hot-path steps on the host (change events)."""
