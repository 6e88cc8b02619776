"""TEST INFRASTRUCTURE: pure-Python restatements of the reference's host-side
algorithms on the hot path (small inputs only).

slice_into_video_intervals follows hwang/video_index.cpp:62-109 statement by
statement, so that the product's re-designed planner can be property-tested
against the reference's control flow on random inputs.  It is pinned by the
golden vectors in tests/golden/host_golden.json (captured from the reference
binary)."""


def slice_into_video_intervals(sample_offsets, sample_sizes, keyframe_indices, rows):
    keyframe_positions = list(keyframe_indices) + [len(sample_sizes)]  # video_index.cpp:64-65
    intervals = []
    start_keyframe_index = 0
    end_keyframe_index = 1
    next_keyframe = keyframe_positions[end_keyframe_index]
    valid_frames = []
    for row in rows:                                                   # :73
        if row >= next_keyframe:                                       # :74
            last_endpoint = sample_offsets[next_keyframe - 1] + sample_sizes[next_keyframe - 1]
            is_adjacent = last_endpoint == sample_offsets[next_keyframe]   # :76-79
            end_keyframe_index += 1
            next_keyframe = keyframe_positions[end_keyframe_index]     # :82
            if row >= next_keyframe or not is_adjacent:                # :84
                if valid_frames:                                       # :86-91
                    intervals.append(((keyframe_positions[start_keyframe_index],
                                       keyframe_positions[end_keyframe_index - 1]),
                                      list(valid_frames)))
                while row >= keyframe_positions[end_keyframe_index]:   # :93-96
                    end_keyframe_index += 1
                valid_frames = []
                start_keyframe_index = end_keyframe_index - 1          # :98
                next_keyframe = keyframe_positions[end_keyframe_index]
        valid_frames.append(row)                                       # :102
    intervals.append(((keyframe_positions[start_keyframe_index],
                       keyframe_positions[end_keyframe_index]), list(valid_frames)))  # :104-107
    return intervals
