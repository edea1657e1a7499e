"""A minimal WebSocket front end for ``StreamServer``: the transport seam of SURVEY.md §8f rank 4.

The reference serves ONE client per process: ``WebSocketAudioSource`` (``sources.py:204-271``) runs a
``websocket_server`` whose text messages are base64-encoded float32 samples (``utils.py:56-66``,
``console/client.py:23``), and ``console/serve.py:119-124`` sends every chunk's ``Annotation.to_rttm()``
back as a text message (only when it is not empty, ``sources.py:270``).  This module speaks the same
messages for MANY clients at once: every connection is one stream of a ``StreamServer`` (the request
path is its id: ``ws://host:port/meeting-7``), audio messages go to ``push()``, a worker thread runs
``step()`` — one GPU batch over all the streams that have a window ready — and each stream gets the
RTTM lines of its own new speech turns.  Closing the socket closes the stream.

Only what that exchange needs of RFC 6455 is implemented, on the standard library (no
``websocket-server`` dependency): the HTTP upgrade handshake, text / binary / continuation / ping /
pong / close frames, client-to-server masking, payloads up to ``max_message`` bytes, no extensions,
no TLS (terminate it in front).  Binary messages are accepted as raw little-endian float32 samples
(no base64), which the reference's client does not send.
"""
from __future__ import annotations

import asyncio
import base64
import hashlib
import struct
import threading
import time
from typing import Dict, Hashable, Optional

import numpy as np

from .serve import StreamServer

_GUID = b"258EAFA5-E914-47DA-95CA-C5AB0DC85B11"
_OP_CONT, _OP_TEXT, _OP_BIN, _OP_CLOSE, _OP_PING, _OP_PONG = 0x0, 0x1, 0x2, 0x8, 0x9, 0xA


def accept_key(key: str) -> str:
    """``Sec-WebSocket-Accept`` for a client's ``Sec-WebSocket-Key`` (RFC 6455 §4.2.2)."""
    return base64.b64encode(hashlib.sha1(key.strip().encode("ascii") + _GUID).digest()).decode("ascii")


def encode_frame(opcode: int, payload: bytes, mask: Optional[bytes] = None) -> bytes:
    """One unfragmented frame; ``mask`` (4 bytes) is what a CLIENT must apply, a server sends none."""
    n = len(payload)
    head = bytes([0x80 | opcode])
    m = 0x80 if mask else 0
    if n < 126:
        head += bytes([m | n])
    elif n < 1 << 16:
        head += bytes([m | 126]) + struct.pack("!H", n)
    else:
        head += bytes([m | 127]) + struct.pack("!Q", n)
    if mask:
        body = bytes(b ^ mask[i & 3] for i, b in enumerate(payload)) if n < 4096 else \
            (np.frombuffer(payload, np.uint8) ^ np.resize(np.frombuffer(mask, np.uint8), n)).tobytes()
        return head + mask + body
    return head + payload


def decode_audio(message) -> np.ndarray:
    """Text message: base64 of float32 bytes (``utils.decode_audio``); binary message: the bytes."""
    raw = base64.decodebytes(message.encode("utf-8")) if isinstance(message, str) else bytes(message)
    if len(raw) % 4:
        raise ValueError(f"audio message of {len(raw)} bytes is not a whole number of float32 samples")
    return np.frombuffer(raw, dtype="<f4")


class _ProtocolError(Exception):
    pass


class _PolicyViolation(Exception):
    pass


class WebSocketFrontEnd:
    """``WebSocketFrontEnd(StreamServer(...), port=7007).start()`` — see the module docstring."""

    def __init__(self, server: StreamServer, host: str = "127.0.0.1", port: int = 7007,
                 idle_sleep: float = 0.002, max_message: int = 16 << 20, max_backlog_seconds: float = 30.0,
                 backlog_timeout: float = 10.0):
        self.server, self.host, self.port = server, host, int(port)
        self.idle_sleep, self.max_message = idle_sleep, int(max_message)
        # back-pressure: a connection whose stream holds more than this much unprocessed audio is not
        # read from (TCP flow control then slows the sender); if the backlog has not drained after
        # `backlog_timeout` seconds (stalled worker, or a client far ahead of real time) the
        # connection is closed with 1008 instead of growing the queue without bound
        self.max_backlog_windows = max(1, int(max_backlog_seconds / server.step_seconds))
        self.backlog_timeout = float(backlog_timeout)
        self._loop: Optional[asyncio.AbstractEventLoop] = None
        self._writers: Dict[Hashable, asyncio.StreamWriter] = {}
        self._threads = []
        self._stop = threading.Event()
        self._ready = threading.Event()
        self._anon = 0
        import collections
        self.errors = collections.deque(maxlen=256)   # (stream id, message) of dropped connections / failed steps

    # ------------------------------------------------------------------ life cycle
    def start(self) -> "WebSocketFrontEnd":
        self._threads = [threading.Thread(target=self._run_loop, name="dz-ws-io", daemon=True),
                         threading.Thread(target=self._run_worker, name="dz-ws-worker", daemon=True)]
        for t in self._threads:
            t.start()
        if not self._ready.wait(10):
            raise RuntimeError("websocket front end did not start")
        return self

    def stop(self) -> None:
        self._stop.set()
        if self._loop is not None:
            self._loop.call_soon_threadsafe(self._loop.stop)
        for t in self._threads:
            t.join(5)

    # ------------------------------------------------------------------ the GPU side
    def _run_worker(self) -> None:
        """``step()`` whenever some stream has a block waiting; RTTM lines back to their sockets."""
        while not self._stop.is_set():
            try:
                out = self.server.step()
            except Exception as e:            # a failing step must not take the transport down silently
                self.errors.append((None, repr(e)))
                time.sleep(0.05)
                continue
            if not out:
                time.sleep(self.idle_sleep)
                continue
            for sid, ann in out.items():
                rttm = ann.to_rttm()
                if rttm and self._loop is not None:               # sources.py:270: empty -> nothing sent
                    self._loop.call_soon_threadsafe(self._send_text, sid, rttm)

    def _send_text(self, sid, text: str) -> None:
        w = self._writers.get(sid)
        if w is not None and not w.is_closing():
            w.write(encode_frame(_OP_TEXT, text.encode("utf-8")))

    # ------------------------------------------------------------------ the socket side
    def _run_loop(self) -> None:
        loop = asyncio.new_event_loop()
        asyncio.set_event_loop(loop)
        self._loop = loop
        srv = loop.run_until_complete(asyncio.start_server(self._client, self.host, self.port))
        self.port = srv.sockets[0].getsockname()[1]               # port 0 -> the one the OS picked
        self._ready.set()
        try:
            loop.run_forever()
        finally:
            srv.close()
            loop.run_until_complete(srv.wait_closed())
            for w in list(self._writers.values()):
                w.close()
            loop.close()

    async def _handshake(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter) -> str:
        head = await asyncio.wait_for(reader.readuntil(b"\r\n\r\n"), 10)
        lines = head.decode("latin-1").split("\r\n")
        parts = lines[0].split()
        if len(parts) < 2 or parts[0] != "GET":
            raise _ProtocolError("not a GET request")
        hdr = {k.strip().lower(): v.strip() for k, v in (ln.split(":", 1) for ln in lines[1:] if ":" in ln)}
        if hdr.get("upgrade", "").lower() != "websocket" or "sec-websocket-key" not in hdr:
            writer.write(b"HTTP/1.1 400 Bad Request\r\nConnection: close\r\n\r\n")
            raise _ProtocolError("not a websocket upgrade")
        if hdr.get("sec-websocket-version", "13") != "13":            # RFC 6455 §4.2.2 / §4.4
            writer.write(b"HTTP/1.1 426 Upgrade Required\r\nSec-WebSocket-Version: 13\r\nConnection: close\r\n\r\n")
            raise _ProtocolError("unsupported Sec-WebSocket-Version")
        writer.write(("HTTP/1.1 101 Switching Protocols\r\nUpgrade: websocket\r\nConnection: Upgrade\r\n"
                      f"Sec-WebSocket-Accept: {accept_key(hdr['sec-websocket-key'])}\r\n\r\n").encode("ascii"))
        return parts[1]

    async def _read_frame(self, reader: asyncio.StreamReader):
        b0, b1 = await reader.readexactly(2)
        n = b1 & 0x7F
        if b0 & 0x70:
            raise _ProtocolError("RSV bits set but no extension was negotiated")   # RFC 6455 §5.2
        if b0 & 0x08 and (not b0 & 0x80 or n > 125):
            raise _ProtocolError("control frames must be unfragmented and at most 125 bytes")   # §5.5
        if n == 126:
            n, = struct.unpack("!H", await reader.readexactly(2))
        elif n == 127:
            n, = struct.unpack("!Q", await reader.readexactly(8))
        if n > self.max_message:
            raise _ProtocolError(f"frame of {n} bytes exceeds max_message")
        if not b1 & 0x80:
            raise _ProtocolError("client frames must be masked")      # RFC 6455 §5.1
        mask = await reader.readexactly(4)
        data = await reader.readexactly(n)
        if n:
            data = (np.frombuffer(data, np.uint8) ^ np.resize(np.frombuffer(mask, np.uint8), n)).tobytes()
        return bool(b0 & 0x80), b0 & 0x0F, data

    @staticmethod
    def _close_with(writer: asyncio.StreamWriter, code: int, reason: str) -> None:
        """A close frame (status code + reason) before the socket goes away (RFC 6455 §7.1.6);
        harmless when the handshake never completed (the peer just sees the connection close)."""
        try:
            if not writer.is_closing():
                writer.write(encode_frame(_OP_CLOSE, struct.pack("!H", code) + reason.encode("utf-8")[:100]))
        except Exception:
            pass

    async def _client(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter) -> None:
        sid, opened = None, False
        try:
            path = await self._handshake(reader, writer)
            sid = path.strip("/") or None
            if sid is None:
                self._anon += 1
                sid = f"stream-{self._anon}"
            # open() resets the slot's clustering / aggregation state under the server lock: off the I/O thread
            await asyncio.get_running_loop().run_in_executor(None, self.server.open, sid)   # ValueError: id in use; RuntimeError: full
            opened = True
            self._writers[sid] = writer
            buf, kind = b"", None
            while True:
                fin, op, data = await self._read_frame(reader)
                if op == _OP_CLOSE:
                    writer.write(encode_frame(_OP_CLOSE, data[:2]))
                    break
                if op == _OP_PING:
                    writer.write(encode_frame(_OP_PONG, data))
                    continue
                if op == _OP_PONG:
                    continue
                if op in (_OP_TEXT, _OP_BIN):
                    buf, kind = data, op
                elif op == _OP_CONT and kind is not None:
                    buf += data
                    if len(buf) > self.max_message:
                        raise _ProtocolError("message exceeds max_message")
                else:
                    raise _ProtocolError(f"unexpected opcode {op}")
                if fin:
                    samples = decode_audio(buf.decode("utf-8") if kind == _OP_TEXT else buf)
                    pending = self.server.push(sid, samples)
                    buf, kind = b"", None
                    waited = 0.0
                    while pending > self.max_backlog_windows:          # stop reading: back-pressure
                        if waited >= self.backlog_timeout:
                            raise _PolicyViolation(f"backlog of {pending} windows did not drain")
                        await asyncio.sleep(0.02)
                        waited += 0.02
                        pending = self.server.pending(sid)
        except (asyncio.IncompleteReadError, ConnectionError):
            pass                                                     # the peer went away
        except _PolicyViolation as e:
            self.errors.append((sid, repr(e)))
            self._close_with(writer, 1008, str(e))
        except _ProtocolError as e:
            self.errors.append((sid, repr(e)))
            self._close_with(writer, 1009 if "max_message" in str(e) else 1002, str(e))
        except Exception as e:
            self.errors.append((sid, repr(e)))
            self._close_with(writer, 1011, "internal error")
        finally:
            if opened:                      # (a refused duplicate id must not close the other connection's stream)
                self._writers.pop(sid, None)
                try:
                    self.server.close(sid)
                except KeyError:
                    pass
            writer.close()
